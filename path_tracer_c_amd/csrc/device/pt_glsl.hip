// pt_glsl.hip -- PTGPU_FLAG_GLSL_SEMANTICS: the reference's OTHER renderer.
//
// The reference holds two statements of its renderer: the CPU one this
// library is built around (src/renderer.c, src/ray.c) and an OpenGL fragment
// shader, shaders/rtx_frag.glsl, which is what its HEAD actually runs
// (SURVEY.md 2.3 lists where they part).  For users who take the shader to
// be "the" renderer this kernel reproduces ITS choices -- not its platform:
// no OpenGL here, one work-item per pixel on gfx950 --
//   rtx_frag.glsl:64-75    PCG hash stream, one state per pixel carried from
//                          sample to sample (so a pixel's samples run in order
//                          on one lane: no sample queue, no lanes-per-pixel)
//   :78-90                 rand_unit_sphere = normalize(3 Box-Muller normals),
//                          six draws per hit
//   :92-117, :121-156      its own quadratic (no unit-direction assumption,
//                          first root >= 0) and Moeller-Trumbore (u, v, t >= 1e-6)
//   :158-192               spheres first, then triangles; the accept rule that
//                          lets a miss overwrite a hit at distance <= 0
//   :194-272               LIFO ray stack, throughput form; roughness not
//                          halved; reflected direction not renormalised, origin
//                          offset 1e-4; the MATERIAL's refractive index; total
//                          internal reflection drops only the transmitted ray
//   :274-295               rays through pixel centres, no per-sample clamp
//   GL_RGB8 store          clamp to [0,1], round to nearest (src/main.c:297,328)
// GLSL leaves the precision of its built-ins open; they are fixed here as
// sequentially rounded binary32 operations (this file is part of the exact
// build: no contraction, IEEE sqrt and divide), the same choice
// oracle/pt_glsl.c makes, with log and cos as the two short kernels of
// include/ptgpu_glsl_math.h (a shared specification, like ptgpu_rng.h), so
// every frame is bit-equal to that oracle's (tested).
#include "pt_launch.h"
#include "pt_trace.h"
#include "pt_vec.h"
#include "ptgpu_glsl_math.h"
#include "ptgpu_rng.h"

namespace pt {
namespace {

constexpr int kGlslStack = kGlslMaxBounces + 2;

// (sqrt_rn / rcp_ieee: pt_vec.h -- the IEEE results of sqrtf() and 1.0f / x in fewer
// instructions, equal on all 2^32 inputs, tools/math_exhaustive.hip)
__device__ __forceinline__ V3 g_normalize(V3 a) { return a * rcp_ieee(sqrt_rn(dot(a, a))); }
__device__ __forceinline__ V3 g_reflect(V3 i, V3 n) { return i - n * (2.0f * dot(n, i)); }
__device__ __forceinline__ V3 g_refract(V3 i, V3 n, float eta) {
    const float ndi = dot(n, i);
    const float k = 1.0f - eta * eta * (1.0f - ndi * ndi);
    if (k < 0.0f)
        return splat(0.0f);
    return i * eta - n * (eta * ndi + sqrt_rn(k));
}

__device__ __forceinline__ float normal_dist(uint32_t &state) {
    const float theta = 2.0f * 3.1415926535897932384626433832795f * ptgpu_pcg_unit(&state);
    const float rho = sqrt_rn(-2.0f * ptgpu_glsl_logf(ptgpu_pcg_unit(&state)));
    return rho * ptgpu_glsl_cosf(theta);
}
__device__ __forceinline__ V3 g_rand_unit_sphere(uint32_t &state) {
    const float x = normal_dist(state);
    const float y = normal_dist(state);
    const float z = normal_dist(state);
    return g_normalize(V3{x, y, z});
}

// :92-117 with o already translated; r2 = radius * radius (uploaded)
__device__ __forceinline__ float g_ray_sphere(V3 o, V3 d, float r2) {
    const float a = dot(d, d);
    const float b = 2.0f * dot(o, d);
    const float c = dot(o, o) - r2;
    const float det = b * b - 4.0f * a * c;
    if (det < 0.0f)
        return -1.0f;
    const float sqrt_det = sqrt_rn(det);
    const float two_a = 2.0f * a;
    const float t1 = (-b - sqrt_det) / two_a;
    if (t1 >= 0.0f)
        return t1;
    const float t2 = (-b + sqrt_det) / two_a;
    if (t2 >= 0.0f)
        return t2;
    return -1.0f;
}

// :121-156 with the uploaded edges e1 = v1 - v0, e2 = v2 - v0
__device__ __forceinline__ float g_ray_triangle(V3 o, V3 d, V3 v0, V3 e1, V3 e2) {
    const float epsilon = 1e-6f;
    const V3 p = cross(d, e2);
    const float det = dot(e1, p);
    if (fabsf(det) < epsilon)
        return -1.0f;
    const float inv_det = 1.0f / det;
    const V3 s = o - v0;
    const float u = inv_det * dot(s, p);
    if (u < epsilon || u > 1.0f)
        return -1.0f;
    const V3 q = cross(s, e1);
    const float v = inv_det * dot(d, q);
    if (v < epsilon || (u + v) > 1.0f)
        return -1.0f;
    const float t = inv_det * dot(e2, q);
    if (t < epsilon)
        return -1.0f;
    return t;
}

struct GHit {
    float distance;
    V3 point, normal;
    uint32_t material;
};

// Point, normal and material of the accepted (distance, object) (:169-176, :182-190).
__device__ __forceinline__ GHit g_surface(const SceneDesc &sc, V3 origin, V3 direction, float distance,
                                          uint32_t object) {
    GHit h;
    h.distance = distance;
    h.point = h.normal = splat(0.0f);
    h.material = 0u;
    if (object != 0xFFFFFFFFu && !(distance < 0.0f)) { // (a miss is never looked at: :206-209)
        const F4 g = sc.geom[object];
        h.point = origin + direction * distance;
        h.material = sc.meta[object] & kMetaMaterialMask;
        if (!(__builtin_bit_cast(uint32_t, g.w) & 0x80000000u)) {
            h.normal = g_normalize(h.point - V3{g.x, g.y, g.z});
        } else {
            const F4 nn = sc.tri[4 * sc.link[object] + 3]; // normalize(cross(v0 - v1, v0 - v2)), uploaded
            const V3 normal = V3{nn.x, nn.y, nn.z};
            const float side = dot(direction, normal) < 0.0f ? 1.0f : -1.0f;
            h.normal = normal * side;
        }
    }
    return h;
}

// :158-192.  The scene records keep the scene's object order; the shader's
// two arrays are that order filtered by kind, hence the two passes.
// The shader rewrites point, normal and material at every candidate it accepts;
// all three are functions of the accepted (distance, object) alone, so the
// scan keeps only that pair and the record is filled in once, after the loop,
// with the very same operations: same bits, a normalisation per ray instead
// of one per accepted candidate.
__device__ __forceinline__ GHit g_trace(const SceneDesc &sc, V3 origin, V3 direction) {
    float distance = -1.0f;
    uint32_t object = 0xFFFFFFFFu; // nothing accepted yet
    for (uint32_t i = 0; i < sc.n_obj; ++i) {
        const F4 g = sc.geom[i];
        if (__builtin_bit_cast(uint32_t, g.w) & 0x80000000u)
            continue; // a triangle record
        const V3 c = V3{g.x, g.y, g.z};
        const float dist = g_ray_sphere(origin - c, direction, g.w);
        if (distance > 0.0f && (dist < 0.0f || dist >= distance))
            continue;
        distance = dist;
        object = i;
    }
    if (sc.n_tri != 0u) {
        for (uint32_t i = 0; i < sc.n_obj; ++i) {
            const F4 g = sc.geom[i];
            if (!(__builtin_bit_cast(uint32_t, g.w) & 0x80000000u))
                continue;
            const uint32_t k = sc.link[i];
            const F4 v0 = sc.tri[4 * k + 0], e1 = sc.tri[4 * k + 1], e2 = sc.tri[4 * k + 2];
            const float dist = g_ray_triangle(origin, direction, V3{v0.x, v0.y, v0.z},
                                              V3{e1.x, e1.y, e1.z}, V3{e2.x, e2.y, e2.z});
            if (distance > 0.0f && (dist < 0.0f || dist >= distance))
                continue;
            distance = dist;
            object = i;
        }
    }
    return g_surface(sc, origin, direction, distance, object);
}

// ---------------------------------------------------------------------------
// The same closest hit through the hierarchy (round 3; scenes of >= 64 objects).
//
// The shader's accept rule (:166, :179) walks its two arrays in order:
//     skip candidate i if (distance > 0 && (dist_i < 0 || dist_i >= distance)),
//     else take it -- a MISS included.
// Working it through: call a test result x *degenerate* if it is +-0 or NaN
// (neither > 0 nor < 0).  A degenerate result is always taken; while the
// running distance is not > 0 everything is taken, misses too; once it is
// > 0 the rule is a running strict minimum over results >= 0.  So with L the
// degenerate result latest in scan order (spheres in object order, then
// triangles), the scan returns the smallest positive result after L, earliest
// position among equals; if there is none: result L itself when L is the very
// last object scanned, and a miss otherwise (the misses after L overwrote it).
// A function of the SET of objects whose test does not return a miss -- which
// the hierarchy may deliver in any order.  As in pt_trace.h, closest_hit_bvh: a
// lane tracks (best positive, earliest position) and the latest degenerate; if
// the best positive lies before L it walks once more for positions after L.
//
// What differs from the default mode's walk is how far the boxes can be
// trusted.  The shader's sphere test is the plain quadratic, whose computed
// discriminant is off by up to ~50 eps (D^2 + r^2) for a sphere of radius r at
// distance D -- "hits" up to sqrt(r^2 + 12 eps (D^2 + r^2)) - r beside the
// sphere, at a reported distance up to ~1e-3 sqrt(D^2 + r^2) off.  So the walk
// pads every box by the largest such margin the ray can meet (GlslCull: extent
// and radii of the scene without the few spheres that dwarf it; those are
// tested for every ray), prunes by the best distance only beyond that noise,
// and screens spheres with the margin the default mode uses (2^-16 of
// D^2 + r^2, 20 x the bound above).  Its triangle test accepts a subset of what
// the default mode's accepts (u, v >= 1e-6 instead of >= 0): same boxes.
// ---------------------------------------------------------------------------
struct GBest {
    float best;        // smallest positive result so far
    uint32_t best_pos; // its scan position + 1 (earliest among equals), 0: none
    float deg_dist;    // the degenerate result latest in scan order ...
    uint32_t deg_pos;  // ... its scan position + 1, 0: none
};

__device__ __forceinline__ float g_test_object(const SceneDesc &sc, V3 o, V3 d, uint32_t i, const F4 g) {
    if (!(__builtin_bit_cast(uint32_t, g.w) & 0x80000000u))
        return g_ray_sphere(o - V3{g.x, g.y, g.z}, d, g.w);
    const uint32_t k = sc.link[i];
    const F4 v0 = sc.tri[4 * k + 0], e1 = sc.tri[4 * k + 1], e2 = sc.tri[4 * k + 2];
    return g_ray_triangle(o, d, V3{v0.x, v0.y, v0.z}, V3{e1.x, e1.y, e1.z}, V3{e2.x, e2.y, e2.z});
}

__device__ __forceinline__ void g_candidate(const SceneDesc &sc, V3 o, V3 d, uint32_t i, const F4 g,
                                            uint32_t after_pos, GBest &u) {
    const bool tri = (__builtin_bit_cast(uint32_t, g.w) & 0x80000000u) != 0u;
    const uint32_t pos = (tri ? sc.n_obj : 0u) + i + 1u;
    if (pos <= after_pos)
        return;
    const float x = g_test_object(sc, o, d, i, g);
    if (x > 0.0f) {
        if (x < u.best || (x == u.best && pos < u.best_pos) || u.best_pos == 0u) {
            u.best = x;
            u.best_pos = pos;
        }
    } else if (!(x < 0.0f)) { // +-0 or NaN
        if (pos > u.deg_pos) {
            u.deg_pos = pos;
            u.deg_dist = x;
        }
    }
}

// conservative "may the shader's test report something other than a miss?" for
// one record, for a direction of any length (inv_dd = 1 / |d|^2)
__device__ __forceinline__ bool g_screen_keep(const F4 g, V3 o, V3 d, float inv_dd) {
    const float r2 = __builtin_fabsf(g.w);
    const float dx = g.x - o.x, dy = g.y - o.y, dz = g.z - o.z;
    const float ddp = __builtin_fmaf(dz, d.z, __builtin_fmaf(dy, d.y, dx * d.x));
    const float dpp = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
    const float perp2 = __builtin_fmaf(-(ddp * inv_dd), ddp, dpp);
    const float excess = __builtin_fmaf(dpp + r2, -PT_SCREEN_MARGIN, perp2 - r2);
    return !(excess > 0.0f);
}

__device__ __forceinline__ void g_walk(const SceneDesc &sc, V3 o, V3 d, float inv_dd, uint32_t *queue,
                                       uint32_t *stack, uint32_t stride, uint32_t after_pos, GBest &u) {
    // ---- what the boxes must be padded by for this ray (see above) -----------
    const GlslCull &gc = sc.glsl;
    float far2 = 0.0f; // squared distance to the farthest corner of the scene's box (without the big spheres)
    {
        const float ax = fmaxf(fabsf(gc.lo[0] - o.x), fabsf(gc.hi[0] - o.x));
        const float ay = fmaxf(fabsf(gc.lo[1] - o.y), fabsf(gc.hi[1] - o.y));
        const float az = fmaxf(fabsf(gc.lo[2] - o.z), fabsf(gc.hi[2] - o.z));
        far2 = ax * ax + ay * ay + az * az;
        if (!(gc.lo[0] <= gc.hi[0]))
            far2 = 0.0f; // nothing but big spheres: no box to pad
    }
    constexpr float kKappa = 0x1.0p-19f; // >= 2.5 x the 12 eps of the bound
    auto margin = [&](float r2) { // how far beside a sphere of this size the test can still "hit"
        const float e = kKappa * (far2 + r2);
        const float by_root = __builtin_sqrtf(e);
        const float by_ratio = e / (2.0f * __builtin_sqrtf(r2));
        return r2 > 0.0f ? fminf(by_root, by_ratio) : by_root;
    };
    float pad = 0.0f, noise = 0.0f;
    if (gc.r2_max > 0.0f || gc.r2_min > 0.0f || far2 > 0.0f) {
        pad = fmaxf(margin(gc.r2_min), margin(gc.r2_max));
        noise = 0x1.0p-9f * __builtin_sqrtf(far2 + gc.r2_max); // reported distance: ~9e-4 sqrt(D^2 + r^2)
    }
#ifdef PT_GLSL_NO_PAD // (experiment: is the padding what keeps the frames equal?  tests say yes)
    pad = noise = 0.0f;
#endif
    const RaySlabs slabs = ray_slabs(o, d);
    const V3 pad_t = V3{pad * fabsf(slabs.inv.x), pad * fabsf(slabs.inv.y), pad * fabsf(slabs.inv.z)};
    auto crosses = [&](const F4 lo, const F4 hi, float &enter) {
        const float x0 = __builtin_fmaf(lo.x, slabs.inv.x, slabs.oi.x), x1 = __builtin_fmaf(hi.x, slabs.inv.x, slabs.oi.x);
        const float y0 = __builtin_fmaf(lo.y, slabs.inv.y, slabs.oi.y), y1 = __builtin_fmaf(hi.y, slabs.inv.y, slabs.oi.y);
        const float z0 = __builtin_fmaf(lo.z, slabs.inv.z, slabs.oi.z), z1 = __builtin_fmaf(hi.z, slabs.inv.z, slabs.oi.z);
        enter = fmaxf(fmaxf(fminf(x0, x1) - pad_t.x, fminf(y0, y1) - pad_t.y), fminf(z0, z1) - pad_t.z);
        const float leave = fminf(fminf(fmaxf(x0, x1) + pad_t.x, fmaxf(y0, y1) + pad_t.y), fmaxf(z0, z1) + pad_t.z);
        const float slack = __builtin_fmaf(PT_BOX_SLACK, fabsf(enter) + fabsf(leave), slabs.slack0);
        const bool missed = (leave + slack < fmaxf(enter, 0.0f)) || (enter - slack - noise > u.best);
        return !missed;
    };
    uint32_t *wr = queue;
    auto resolve = [&]() {
        for (uint32_t *rd = queue; pt_any(rd < wr); rd += stride) {
            if (rd < wr) {
                const uint32_t idx = *rd;
                g_candidate(sc, o, d, idx, sc.geom[idx], after_pos, u);
            }
        }
        wr = queue;
    };
    // objects outside the hierarchy: no finite box, or spheres that dwarf the scene
    for (uint32_t k = 0; k < sc.n_unbounded; ++k) {
        const uint32_t idx = sc.bvh_unbounded[k];
        g_candidate(sc, o, d, idx, sc.geom[idx], after_pos, u);
    }

    constexpr uint32_t kDone = 0xFFFFFFFFu;
    uint32_t cur = sc.bvh_root;
    uint32_t *sp = stack;
    while (pt_any(cur != kDone)) {
        while (pt_any(cur != kDone && !(cur & kLeafRef))) {
            if (cur != kDone && !(cur & kLeafRef)) {
                const F4 a = sc.bvh_nodes[2u * cur + 0u];
                const F4 b = sc.bvh_nodes[2u * cur + 1u];
                const uint32_t a0 = __builtin_bit_cast(uint32_t, a.x), a1 = __builtin_bit_cast(uint32_t, a.y),
                               a2 = __builtin_bit_cast(uint32_t, a.z);
                const uint32_t b0 = __builtin_bit_cast(uint32_t, b.x), b1 = __builtin_bit_cast(uint32_t, b.y),
                               b2 = __builtin_bit_cast(uint32_t, b.z);
                float l_enter, r_enter;
                const bool go_l = crosses(F4{half_lo(a0), half_hi(a0), half_lo(a1), 0.0f},
                                          F4{half_hi(a1), half_lo(a2), half_hi(a2), 0.0f}, l_enter);
                const bool go_r = crosses(F4{half_lo(b0), half_hi(b0), half_lo(b1), 0.0f},
                                          F4{half_hi(b1), half_lo(b2), half_hi(b2), 0.0f}, r_enter);
                const uint32_t l_ref = __builtin_bit_cast(uint32_t, a.w), r_ref = __builtin_bit_cast(uint32_t, b.w);
                const bool both = go_l && go_r;
                const bool none = !go_l && !go_r;
                const bool r_first = go_r && (!go_l || r_enter < l_enter);
                if (both) {
                    *sp = r_first ? l_ref : r_ref;
                    sp += stride;
                }
                cur = r_first ? r_ref : l_ref;
                if (none) {
                    if (sp != stack) {
                        sp -= stride;
                        cur = *sp;
                    } else {
                        cur = kDone;
                    }
                }
            }
        }
        if (cur != kDone) { // a leaf: its records side by side (pt_trace.h, visit_leaf)
            const uint32_t count = (cur >> 28) & 7u;
            const uint32_t at = cur & 0x0FFFFFFFu;
            F4 g[kLeafSlots];
            uint32_t id[kLeafSlots];
#pragma unroll
            for (uint32_t k = 0; k < kLeafSlots; ++k) {
                g[k] = sc.bvh_leaf_geom[at + k];
                id[k] = sc.bvh_order[at + k];
            }
#pragma unroll
            for (uint32_t k = 0; k < kLeafSlots; ++k) {
                if (k < count && g_screen_keep(g[k], o, d, inv_dd)) {
                    *wr = id[k];
                    wr += stride;
                }
            }
            if (sp != stack) {
                sp -= stride;
                cur = *sp;
            } else {
                cur = kDone;
            }
        }
        if (pt_any(wr > queue)) // (as early as possible: the best distance prunes the walk)
            resolve();
    }
    resolve();
}

__device__ __forceinline__ GHit g_trace_bvh(const SceneDesc &sc, V3 o, V3 d, uint32_t *queue,
                                            uint32_t *stack, uint32_t stride) {
    // a ray that is no geometric object -- a NaN or infinite component, a
    // direction of no usable length -- gets the sequential scan
    const float dd = dot(d, d);
    const float probe = (fabsf(o.x) + fabsf(o.y) + fabsf(o.z)) + (fabsf(d.x) + fabsf(d.y) + fabsf(d.z));
    const bool odd = !(probe < __builtin_inff()) || !(dd > 0x1.0p-40f && dd < 0x1.0p40f);
    GBest u = GBest{__builtin_inff(), 0u, 0.0f, 0u};
    const float inv_dd = __builtin_amdgcn_rcpf(dd);
    bool walking = !odd;
    uint32_t after_pos = 0u;
    const uint32_t last = sc.glsl.last_scanned;
    const uint32_t last_pos =
        last == 0xFFFFFFFFu ? 0u
                            : ((__builtin_bit_cast(uint32_t, sc.geom[last].w) & 0x80000000u) ? sc.n_obj : 0u) + last + 1u;
    for (int walk = 0; walk < 2 && pt_any(walking); ++walk) {
        if (walking)
            g_walk(sc, o, d, inv_dd, queue, stack, stride, after_pos, u);
        // the answer lies beyond a degenerate result only if a positive one came before it
        const bool again = walking && walk == 0 && u.deg_pos != 0u && u.best_pos != 0u &&
                           u.best_pos < u.deg_pos && u.deg_pos != last_pos;
        if (again) {
            after_pos = u.deg_pos;
            u.best = __builtin_inff();
            u.best_pos = 0u;
        }
        walking = again;
    }
    float distance = -1.0f;
    uint32_t pos = 0u;
    if (u.best_pos != 0u && u.best_pos > u.deg_pos) {
        distance = u.best;
        pos = u.best_pos;
    } else if (u.deg_pos != 0u && u.deg_pos == last_pos) {
        distance = u.deg_dist;
        pos = u.deg_pos;
    }
    uint32_t object = 0xFFFFFFFFu;
    if (pos != 0u)
        object = pos - 1u >= sc.n_obj ? pos - 1u - sc.n_obj : pos - 1u;
    GHit h = g_surface(sc, o, d, distance, object);
    if (pt_any(odd)) {
        if (odd)
            h = g_trace(sc, o, d);
    }
    return h;
}

struct GItem {
    V3 origin, direction, color;
    int bounces;
};

__device__ __forceinline__ uint8_t unorm8(float v) {
    if (!(v > 0.0f))
        return 0;
    if (v > 1.0f)
        v = 1.0f;
    return (uint8_t)(v * 255.0f + 0.5f);
}

} // namespace

// kBvh: closest hits through the hierarchy (g_trace_bvh) instead of the
// shader's own scan of every object (g_trace); same frame.
template <bool kBvh>
__global__ void __launch_bounds__(64) pt_glsl_kernel(const LaunchArgs a) {
    extern __shared__ uint32_t glsl_lds[]; // kBvh: candidate queues + traversal stacks, one column per lane
    uint32_t *const queue = glsl_lds + threadIdx.x;
    uint32_t *const walk_stack = queue + kScreenQueue * 64u;
    const uint32_t npix = a.num_rows * a.framew;
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = j < npix;
    const uint32_t W = a.framew;
    const uint32_t lrow = (valid ? j : 0u) / W;
    const uint32_t sx = (valid ? j : 0u) - lrow * W;
    uint32_t grow = a.row_begin + lrow;
    if (a.tile_rows != 0u) {
        const uint32_t tile = lrow / a.tile_rows;
        grow = a.row_begin + tile * a.tile_rows * a.tile_stride + (lrow - tile * a.tile_rows);
    }
    const uint64_t out_pixel = (uint64_t)(a.compact ? lrow : grow) * W + sx;

    const float fw = (float)a.framew, fh = (float)a.frameh;
    const float aspect = fw / fh;
    const float cx = ((float)sx + 0.5f) / fw * 2.0f - 1.0f;
    const float cy = 1.0f - ((float)grow + 0.5f) / fh * 2.0f;
    const float fidx = (cy * fh) * fw + (cx * fw);
    uint32_t rng = (uint32_t)(long long)fidx + a.seed;
    const V3 dir0 = g_normalize(V3{cx * a.tan_half_fov, cy * a.tan_half_fov / aspect, 1.0f});
    const V3 sky = V3{a.scene.sky[0], a.scene.sky[1], a.scene.sky[2]};

    GItem stack[kGlslStack];
    V3 result = splat(0.0f);
    uint32_t rays = 0;
    const uint32_t spp = valid ? a.spp : 0u;
    for (uint32_t s = 0; s < spp; ++s) {
        rng += s;
        int n = 0;
        stack[0].origin = splat(0.0f);
        stack[0].direction = dir0;
        stack[0].color = splat(1.0f);
        stack[0].bounces = 0;
        n = 1;
        V3 total = splat(0.0f);
        while (n > 0) {
            GItem cur = stack[--n];
            if ((uint32_t)cur.bounces > a.max_bounces) {
                total = total + cur.color * sky;
                continue;
            }
            ++rays;
            const GHit hit = kBvh ? g_trace_bvh(a.scene, cur.origin, cur.direction, queue, walk_stack, 64u)
                                  : g_trace(a.scene, cur.origin, cur.direction);
            if (hit.distance < 0.0f) {
                total = total + cur.color * sky;
                continue;
            }
            // material: albedo | Le = emission_color * emission_strength (the same
            // product the shader forms, made at upload), transparency | and the two
            // fields only this mode reads: the un-halved roughness and the
            // refractive index (pt_layout.h, mat_extra)
            const F4 m0 = a.scene.mat[2 * hit.material + 0];
            const F4 m1 = a.scene.mat[2 * hit.material + 1];
            const F4 mx = a.scene.mat_extra[hit.material];
            const V3 le = V3{m1.x, m1.y, m1.z};
            total = total + cur.color * le;
            cur.color = cur.color * V3{m0.x, m0.y, m0.z};
            const V3 deviation = g_rand_unit_sphere(rng) * mx.x;
            const V3 normal = g_normalize(hit.normal + deviation);
            const float transparency = m1.w, ior = mx.y;
            if (transparency < 1.0f) {
                GItem next;
                next.direction = g_reflect(cur.direction, normal);
                next.origin = hit.point + next.direction * 0.0001f;
                next.bounces = cur.bounces + 1;
                next.color = cur.color * (1.0f - transparency);
                stack[n++] = next;
            }
            if (transparency > 0.0f) {
                const float facing = dot(cur.direction, normal);
                const float eta = facing < 0.0f ? 1.0f / ior : ior;
                const V3 rn = facing < 0.0f ? normal : normal * -1.0f;
                GItem next;
                next.direction = g_refract(cur.direction, rn, eta);
                if (next.direction.x == 0.0f && next.direction.y == 0.0f && next.direction.z == 0.0f)
                    continue;
                next.origin = hit.point + next.direction * 0.0001f;
                next.bounces = cur.bounces + 1;
                next.color = cur.color * transparency;
                stack[n++] = next;
            }
        }
        result = result + total;
    }
    if (valid) {
        const float n = (float)a.spp;
        uint8_t *dst = a.frame + 3u * out_pixel;
        dst[0] = unorm8(result.x / n);
        dst[1] = unorm8(result.y / n);
        dst[2] = unorm8(result.z / n);
    }
    if (a.counters != nullptr) {
        // one pair of atomics per wave (this mode is not the tuned path)
        uint32_t r = rays, sm = spp;
        for (int off = 32; off > 0; off >>= 1) {
            r += __shfl_xor(r, off, 64);
            sm += __shfl_xor(sm, off, 64);
        }
        if ((threadIdx.x & 63u) == 0u) {
            Counters *c = a.counters + (blockIdx.x & (kCounterShards - 1u));
            atomicAdd(&c->rays, (unsigned long long)r);
            atomicAdd(&c->scans, (unsigned long long)r);
            atomicAdd(&c->samples, (unsigned long long)sm);
        }
    }
}

hipError_t launch_glsl(const LaunchArgs &a, hipStream_t stream) {
    const uint64_t npix = (uint64_t)a.num_rows * a.framew;
    if (npix == 0 || a.spp == 0)
        return hipSuccess;
    const uint32_t block = 64;
    const uint32_t grid = (uint32_t)((npix + block - 1) / block);
    // (this mode has two scan forms: the shader's own and the hierarchy; a scene whose
    // default-mode choice is the screened linear scan takes the hierarchy if it has one)
    if (a.screened != 0u && a.scene.bvh_nodes != nullptr) {
        const size_t lds = (size_t)4 * 64u * (kScreenQueue + a.scene.bvh_depth + 2u);
        if (lds > 64u * 1024u) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(pt_glsl_kernel<true>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess)
                return e;
        }
        hipLaunchKernelGGL(pt_glsl_kernel<true>, dim3(grid), dim3(block), lds, stream, a);
    } else {
        hipLaunchKernelGGL(pt_glsl_kernel<false>, dim3(grid), dim3(block), 0, stream, a);
    }
    return hipGetLastError();
}

} // namespace pt
