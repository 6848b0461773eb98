// pt_shade.h -- one recursion level of incident_light (src/renderer.c:21-74)
// for one lane, and what a path carries from level to level.
//
// incident_light's recursion is evaluated as a forward walk that records one
// 16-bit material index per bounce in an LDS stack (advance_path), and an
// unwind that re-applies `Le + lerp(refl, trans, t) * albedo` from the deepest
// level up (unwind_path) -- the reference's own evaluation order, so results
// are bit-equal to a sequential evaluation, which a throughput-product
// formulation is not.  Per-lane xorshift32 in a register, re-seeded per
// (pixel, sample) (include/ptgpu_rng.h), three draws per hit like
// rand_unit_sphere (src/renderer.c:11-19).
#pragma once

#include "pt_layout.h"
#include "pt_trace.h"
#include "pt_vec.h"
#include "ptgpu_rng.h"

namespace PT_NS {


// Refraction constants of src/renderer.c:58-63, with the reference's types:
//   float refractive_index = dot < 0 ? 1.5 : 1 / 1.5;     (double -> float)
//   eta = 1.0 / refractive_index                          (double -> float)
__device__ constexpr float kIorEnter = (float)1.5;
__device__ constexpr float kIorLeave = (float)(1 / 1.5);
__device__ constexpr float kEtaEnter = (float)(1.0 / (double)kIorEnter);
__device__ constexpr float kEtaLeave = (float)(1.0 / (double)kIorLeave);
__device__ constexpr float kRayOffset = (float)0.001; // src/renderer.c:49,66

// rand_unit_sphere (src/renderer.c:11-19): normalise a point of the cube
// [-1,1)^3; three draws in x, y, z order.
//
// u = (x >> 8) * 2^-24 and u * 2 - 1 (src/renderer.c:12-14 through
// include/ptgpu_rng.h) are three operations of which two are EXACT -- a 24-bit
// integer scaled by a power of two, then doubled -- so the value is the one
// rounding of k * 2^-23 - 1, and one fused multiply-add computes exactly that:
// the same bits as the sequential form in both builds (and as the oracle's),
// two instructions less per component.
__device__ __forceinline__ float unit_cube_component(uint32_t x) {
    return __builtin_fmaf((float)(x >> 8), 0x1.0p-23f, -1.0f);
}
__device__ __forceinline__ V3 rand_unit_sphere(uint32_t &state) {
    state = ptgpu_xorshift32(state);
    const float cx = unit_cube_component(state);
    state = ptgpu_xorshift32(state);
    const float cy = unit_cube_component(state);
    state = ptgpu_xorshift32(state);
    const float cz = unit_cube_component(state);
    return normalize(V3{cx, cy, cz});
}

// What one surface interaction decides (src/renderer.c:34-70), given the
// perturbed normal: the next ray, or that the path has no child at this
// level (total internal reflection / NaN transparency).
struct Bounce {
    V3 origin, direction;
    bool has_child;
};

// origin of a child ray: point + direction * 0.001 (src/renderer.c:49,66)
__device__ __forceinline__ V3 ray_origin(V3 point, V3 direction) {
#if defined(PT_FUSE)
    return madd(direction, kRayOffset, point);
#else
    return point + direction * kRayOffset;
#endif
}

__device__ __forceinline__ Bounce reflect_ray(V3 d, V3 n, V3 point) {
    Bounce b;
    b.direction = normalize_dir(reflect(d, n));
    b.origin = ray_origin(point, b.direction);
    b.has_child = true;
    return b;
}

__device__ __forceinline__ Bounce refract_ray(V3 d, V3 n, V3 point) {
    Bounce b;
    float facing = dot(d, n);
    float eta = facing < 0.0f ? kEtaEnter : kEtaLeave;
    V3 rn = facing < 0.0f ? n : -n;
    V3 bent;
    b.has_child = refract(d, rn, eta, bent);
    b.direction = normalize_dir(bent);
    b.origin = ray_origin(point, b.direction);
    return b;
}

// One level of the unwind: Le + lerp(reflected, transmitted, t) * albedo
// (src/renderer.c:72-73) where exactly one of the two children is `child`
// and the other is the zero vector.
__device__ __forceinline__ V3 shade_level(const F4 m0, const F4 m1, V3 child) {
    const float t = m1.w;
    const bool via_reflection = t < 1.0f;
    V3 refl = via_reflection ? child : splat(0.0f);
    V3 trans = via_reflection ? splat(0.0f) : child;
    V3 li = lerp(refl, trans, t);
#if defined(PT_FUSE)
    return madd(li, xyz(m0), xyz(m1));
#else
    return xyz(m1) + li * xyz(m0);
#endif
}

// ---------------------------------------------------------------------------
// Branching materials (0 < transparency < 1): incident_light calls itself
// twice at such a hit, for the reflected and then for the transmitted ray
// (src/renderer.c:45-52, 56-70), and a sample is a depth-first walk of a binary
// tree.  A lane walks it as a path that can be RESUMED: at a branching hit it
// follows the reflection and puts the transmitted ray aside (both are computed
// from the same perturbed normal, at the hit); when the path ends, the unwind
// (settle_path) returns values upwards only as far as the deepest level whose
// transmitted ray is still waiting, keeps that level's reflected_Li in the
// ray's place and sends the lane off along the transmitted ray; the next time
// the unwind passes the level it has both children.  Draw order and arithmetic
// are those of the recursion.  Everything else -- schedules, scans, the
// 16-bit material stack -- is shared with the non-branching paths, which are
// the special case of no level ever waiting.
//
// Per lane: two bit masks (bit k = recursion level k) and six words per level
// in private memory, touched at branching hits only.
// ---------------------------------------------------------------------------
constexpr int kTreeFrames = 16; // >= PTGPU_MAX_BOUNCES_TREE + 1 levels

template <bool kTree> struct Branches {};
template <> struct Branches<true> {
    uint32_t waiting = 0;  // level k's transmitted ray is in put_aside[k], not traced yet
    uint32_t returned = 0; // level k's reflected_Li is in put_aside[k][0..2], its
                           // transmission child is being evaluated
    float put_aside[kTreeFrames][6];
};

// Le + lerp(reflected, transmitted, t) * albedo with both children at hand
// (src/renderer.c:72-73)
__device__ __forceinline__ V3 shade_level_both(const F4 m0, const F4 m1, V3 refl, V3 trans) {
    V3 li = lerp(refl, trans, m1.w);
#if defined(PT_FUSE)
    return madd(li, xyz(m0), xyz(m1));
#else
    return xyz(m1) + li * xyz(m0);
#endif
}

// ---------------------------------------------------------------------------
// Wave-level schedule.  Each pass of the loop has two stages:
//   trace+shade  lanes that own a live ray scan the scene and shade the hit,
//                which either yields the next ray or ends the path (the lane
//                then PARKS with the value its deepest call returns);
//   settle       when the ballot says every lane is parked, or at least
//                `settle_lanes` of them are, all parked lanes together unwind
//                their recursion, add the clamped sample and regenerate their
//                next sample's primary ray.
// settle_lanes = 1 is plain per-lane path regeneration; 64 is a sample-
// synchronous wave.  The settle stage is the divergent, scene-independent part
// of a pass (9 dependent LDS-gather levels at 8 bounces); batching it trades a
// few idle lane-passes for running it with many lanes instead of one or two.
// The host picks settle_lanes from the scene size (a pass over 1000 objects
// dwarfs a settle; a pass over 8 does not).  The image does not depend on it.
// ---------------------------------------------------------------------------

// Diagnostic build only (make EXTRA_HIPFLAGS=-DPT_STAMPS): wave-level cycle
// shares of the scene scan, the shading and the settle stage, summed into
// three device words that no kernel reads and no output depends on
// (tools/stamps.py prints them).  The shipped library contains no stamp.
#ifdef PT_STAMPS
#define PT_STAMP(var)                                                          \
    unsigned long long var;                                                    \
    __builtin_amdgcn_sched_barrier(0);                                         \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory");  \
    __builtin_amdgcn_sched_barrier(0)
#define PT_STAMP_ADD(field, later, earlier) (field) += (later) - (earlier)
__device__ unsigned long long g_stamps[8];
#else
#define PT_STAMP(var)
#define PT_STAMP_ADD(field, later, earlier)
#endif

// State of one lane's path.
struct PathState {
#ifdef PT_STAMPS
    unsigned long long t_scan = 0, t_shade = 0, t_settle = 0, t_total = 0;
#endif
    V3 o, d;        // ray to trace next
    V3 r;           // value returned by the deepest call, once parked
    uint32_t depth; // hits recorded on the bounce stack so far
    uint32_t rng;
    bool live;      // owns a ray to trace
    bool parked;    // path ended, waiting for the settle stage
};

// Diagnostic build only (make EXTRA_HIPFLAGS=-DPT_SCAN_AUDIT; tools/inplane_audit.py): every ray a
// culling scan (screened / hierarchy) has answered is put through the plain scan of src/ray.c:12-65
// as well, and for every triangle the conservative screen would cull the audit asks how close the
// ray was to the regime in which the reference's triangle test stops being geometry (DESIGN.md
// 4.1, "in-plane rays"):
//   [0] rays audited                      [1] rays whose culling-scan hit differs from the plain scan's
//   [2] (ray, triangle) pairs the screen culls
//   [3] ... of those, ray within 1e-3 rad of the triangle's plane and passing within 3 bounding radii
//   [4] ... of those culled pairs, the reference's own test reports a hit (noise hit, culled)
//   [5] ... and that hit is nearer than the ray's accepted hit (it would have changed the answer)
// Summed into device words no kernel reads.  Never in the shipped library.
#if defined(PT_SCAN_AUDIT) && !defined(PT_FAST_MATH)
__device__ unsigned long long g_scan_audit[8];
__device__ __forceinline__ void audit_add(int slot, bool flag) {
    const unsigned long long m = pt_ballot(flag);
    if (m != 0ull && (threadIdx.x & 63u) == (uint32_t)(__ffsll((long long)pt_ballot(true)) - 1))
        atomicAdd(&g_scan_audit[slot], (unsigned long long)__popcll(m));
}
__device__ __forceinline__ void audit_ray(const SceneRecords &rec, V3 o, V3 d, Hit got, bool active) {
    if (!pt_any(active))
        return;
    Hit plain = Hit{-1.0f, -1};
    if (active)
        plain = closest_hit(rec, o, d);
    audit_add(0, active);
    audit_add(1, active && (plain.object != got.object ||
                            __builtin_bit_cast(uint32_t, plain.distance) != __builtin_bit_cast(uint32_t, got.distance)));
    uint32_t culled = 0, inplane = 0, noise = 0, nearer = 0;
    if (active && !direction_is_not_unit(d)) {
        for (uint32_t i = 0; i < rec.n_obj; ++i) {
            const F4 g = rec.geom[i];
            if (!is_triangle_record(g) || screen_keep(g, o, d))
                continue;
            ++culled;
            const uint32_t k = rec.link[i];
            const V3 n = xyz(rec.tri[4 * k + 3]);
            const V3 dp = xyz(g) - o;
            const float along = dot(dp, d);
            const V3 off = dp - d * along;
            const float r2 = __builtin_fabsf(g.w);
            if (__builtin_fabsf(dot(n, d)) < 1.0e-3f && dot(off, off) <= 9.0f * r2)
                ++inplane;
            float dist = 0.0f;
            if (ray_triangle(o, d, xyz(rec.tri[4 * k + 0]), xyz(rec.tri[4 * k + 1]), xyz(rec.tri[4 * k + 2]), dist)) {
                ++noise;
                if (got.object < 0 || dist < got.distance)
                    ++nearer;
            }
        }
    }
    // (called from divergent code: every lane adds for itself)
    if (culled)
        atomicAdd(&g_scan_audit[2], (unsigned long long)culled);
    if (inplane)
        atomicAdd(&g_scan_audit[3], (unsigned long long)inplane);
    if (noise)
        atomicAdd(&g_scan_audit[4], (unsigned long long)noise);
    if (nearer)
        atomicAdd(&g_scan_audit[5], (unsigned long long)nearer);
}
#endif

// kScan: 0 plain linear scan, 1 screened linear scan, 2 BVH scan.
template <int kScan>
__device__ __forceinline__ Hit scan_scene(const SceneRecords &rec, V3 o, V3 d,
                                          const ScreenQueue queue) {
    if (kScan == 2)
        return closest_hit_bvh(rec, o, d, queue);
    if (kScan == 1)
        return closest_hit_screened(rec, o, d, queue);
    return closest_hit(rec, o, d);
}

// One recursion level of incident_light for a live lane: scan the scene, then
// either end the path (miss, depth limit, total internal reflection) or
// produce the next ray (src/renderer.c:24-70).
// `known` (wave-uniform): the closest hit of p's ray is already in `h_known`.
template <int kScan, bool kTree>
__device__ __forceinline__ void
advance_path(const SceneRecords &rec, const LaunchArgs &a, uint16_t *stack,
             uint32_t stack_stride, const ScreenQueue queue, V3 sky,
             PathState &p, Branches<kTree> &br, bool known = false,
             Hit h_known = Hit{-1.0f, -1}) {
    PT_STAMP(stamp0);
    const Hit h = known ? h_known : scan_scene<kScan>(rec, p.o, p.d, queue);
#if defined(PT_SCAN_AUDIT) && !defined(PT_FAST_MATH)
    if (kScan != 0) // (all lanes that are here hold a live ray)
        audit_ray(rec, p.o, p.d, h, true);
#endif
    PT_STAMP(stamp1);
    PT_STAMP_ADD(p.t_scan, stamp1, stamp0);
    if (h.distance < 0.0f) { // miss: this level returns the sky colour
        p.r = sky;
        p.live = false;
    } else {
        const Surface sf = surface_at(rec, p.o, p.d, h);
        stack[p.depth * stack_stride] = (uint16_t)sf.material;
        ++p.depth;
        const F4 m0 = rec.mat[2 * sf.material + 0];
        const float transparency = rec.mat[2 * sf.material + 1].w;

        V3 deviation = rand_unit_sphere(p.rng) * m0.w; // roughness * 0.5
        V3 n = normalize(sf.normal + deviation);

        Bounce b;
        if (transparency < 1.0f) {
            if constexpr (kTree) {
                if (transparency > 0.0f) { // both children: the transmitted ray waits
                    const Bounce later = refract_ray(p.d, n, sf.point);
                    if (later.has_child) { // (total internal reflection: transmitted_Li = 0)
                        const uint32_t k = p.depth - 1u;
                        br.put_aside[k][0] = later.origin.x;
                        br.put_aside[k][1] = later.origin.y;
                        br.put_aside[k][2] = later.origin.z;
                        br.put_aside[k][3] = later.direction.x;
                        br.put_aside[k][4] = later.direction.y;
                        br.put_aside[k][5] = later.direction.z;
                        br.waiting |= 1u << k;
                    }
                }
            }
            b = reflect_ray(p.d, n, sf.point);
        } else if (transparency > 0.0f) {
            b = refract_ray(p.d, n, sf.point);
        } else { // NaN transparency: neither branch is taken
            b.has_child = false;
            b.origin = p.o;
            b.direction = p.d;
        }
        p.o = b.origin;
        p.d = b.direction;
        p.r = splat(0.0f);
        p.live = b.has_child;
        if (p.live && p.depth > a.max_bounces) {
            // the child call has bounces > MAX_BOUNCES: sky, no trace
            p.r = sky;
            p.live = false;
        }
    }
    p.parked = !p.live;
    PT_STAMP(stamp2);
    PT_STAMP_ADD(p.t_shade, stamp2, stamp1);
}

// The unwind of a parked lane: re-apply Le + lerp(refl, trans, t) * albedo
// from the deepest recorded level up; returns the clamped sample
// (src/renderer.c:72-73, 88).
__device__ __forceinline__ V3 unwind_path(const SceneRecords &rec,
                                          const uint16_t *stack,
                                          uint32_t stack_stride,
                                          const PathState &p) {
    V3 r = p.r;
    for (uint32_t k = p.depth; k-- > 0;) {
        const uint32_t m = stack[k * stack_stride];
        r = shade_level(rec.mat[2 * m + 0], rec.mat[2 * m + 1], r);
    }
    return clamp01(r);
}

// The unwind of a parked lane in a scene with branching materials: as
// unwind_path, but only as far as the deepest level whose transmitted ray still
// waits.  Returns true when the sample is complete (clamped value in `sample`),
// false when the lane's walk goes on: p holds the transmitted ray and is live.
template <bool kTree>
__device__ __forceinline__ bool settle_path(const SceneRecords &rec, const LaunchArgs &a,
                                            const uint16_t *stack, uint32_t stack_stride,
                                            V3 sky, PathState &p, Branches<kTree> &br,
                                            V3 &sample) {
    if constexpr (!kTree) {
        sample = unwind_path(rec, stack, stack_stride, p);
        return true;
    } else {
        V3 r = p.r;
        for (uint32_t k = p.depth; k-- > 0;) {
            const uint32_t m = stack[k * stack_stride];
            const F4 m0 = rec.mat[2 * m + 0], m1 = rec.mat[2 * m + 1];
            const uint32_t bit = 1u << k;
            if (br.waiting & bit) {
                // r is level k's reflected_Li; its transmission child comes now
                br.waiting &= ~bit;
                const V3 o = V3{br.put_aside[k][0], br.put_aside[k][1], br.put_aside[k][2]};
                const V3 d = V3{br.put_aside[k][3], br.put_aside[k][4], br.put_aside[k][5]};
                if (k + 1u > a.max_bounces) {
                    // that call has bounces > MAX_BOUNCES: sky, no ray (src/renderer.c:24-26)
                    r = shade_level_both(m0, m1, r, sky);
                    continue;
                }
                br.put_aside[k][0] = r.x;
                br.put_aside[k][1] = r.y;
                br.put_aside[k][2] = r.z;
                br.returned |= bit;
                p.o = o;
                p.d = d;
                p.r = splat(0.0f);
                p.depth = k + 1u;
                p.live = true;
                p.parked = false;
                return false;
            }
            if (br.returned & bit) { // r is level k's transmitted_Li
                br.returned &= ~bit;
                const V3 refl = V3{br.put_aside[k][0], br.put_aside[k][1], br.put_aside[k][2]};
                r = shade_level_both(m0, m1, refl, r);
                continue;
            }
            r = shade_level(m0, m1, r);
        }
        sample = clamp01(r);
        return true;
    }
}

// Wave-uniform decision of the settle stage (see run_paths).
__device__ __forceinline__ bool settle_now(bool live, bool parked,
                                           uint32_t settle_lanes) {
    const unsigned long long parked_mask = pt_ballot(parked);
    const unsigned long long live_mask = pt_ballot(live);
    return parked_mask != 0ull &&
           (live_mask == 0ull || (uint32_t)__popcll(parked_mask) >= settle_lanes);
}

} // namespace PT_NS
