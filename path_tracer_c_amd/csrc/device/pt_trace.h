// pt_trace.h -- closest-hit query of one ray against the whole scene.
//
// Device counterpart of the reference's trace_ray (src/ray.c:4-73) with the
// two cglm intersection routines it calls (glms_ray_sphere src/ray.c:21,
// glms_ray_triangle src/ray.c:42-43; published definitions, SURVEY.md
// section 8(c)).  The object loop is uniform across a wavefront (every lane
// walks the same object i), so `rec.geom(i)` / `rec.meta(i)` are broadcast
// LDS reads (or scalar loads in the global-scene variant) and only the
// per-lane accept/reject is divergent.
//
// The reference stores point/normal/material on every accepted candidate;
// here only (distance, object index) are tracked inside the loop and the
// point and normal are computed once afterwards from the same inputs -- the
// same values, since both are pure functions of (ray, distance, object).
#pragma once

#include "pt_layout.h"
#include "pt_vec.h"

namespace PT_NS {

// A view of the scene records, in LDS or in global memory.
struct SceneRecords {
    const F4 *geom;
    const F4 *tri;
    const F4 *mat;
    const uint32_t *meta;
    const uint32_t *link;
    uint32_t n_obj;
    // hierarchy (always in global memory), see csrc/bvh_build.hpp
    const F4 *bvh_nodes;
    const uint32_t *bvh_order;
    const F4 *bvh_leaf_geom;
    const uint32_t *bvh_unbounded;
    uint32_t n_nodes;
    uint32_t n_unbounded;
    uint32_t bvh_root;
    uint32_t last_sphere; // highest sphere index, 0xFFFFFFFF: none
};

PT_HD V3 xyz(const F4 &r) { return V3{r.x, r.y, r.z}; }

// One 16-byte fetch of a record a LANE picks by itself (a queued candidate, a
// hit object).  Read field by field, the compiler fetched .w first -- sphere or
// triangle? -- and the centre only inside the sphere branch: two memory round
// trips in a row where one brings the whole record.
typedef float pt_float4 __attribute__((ext_vector_type(4)));
PT_HD F4 load_record(const F4 *p) {
    const pt_float4 v = *reinterpret_cast<const pt_float4 *>(p);
    return F4{v.x, v.y, v.z, v.w};
}

// geom[i].w is r*r (sign bit clear) for a sphere and -(bounding radius)^2 for
// a triangle: the sign bit tells the two apart without a second load, and
// |w| is a squared radius the screen can use for either.
PT_HD bool is_triangle_record(const F4 &g) {
    return (__builtin_bit_cast(uint32_t, g.w) & 0x80000000u) != 0u;
}

// glms_ray_sphere for a unit direction; `r2` is radius*radius.
// Returns false when there is no root or both roots are behind the origin.
// `dist`: the root trace_ray goes on with, `t1 > 0 ? t1 : t2` (src/ray.c:24) -- written
// only where the roots exist (the caller initialises it), so lanes without a real root
// need neither root set to anything.
PT_HD bool ray_sphere(V3 o, V3 d, V3 c, float r2, float &dist) {
#if defined(__HIP_DEVICE_COMPILE__) && defined(PT_FAST_SPHERE) && !defined(PT_X_SPHERE)
    // Tolerance mode: the reference's formulation -- remedy term, roots as q and
    // c/q -- with fused multiply-adds and v_sqrt_f32 / v_rcp_f32 as they come.
    // (The 17-FLOP discriminant b^2 - (|dp|^2 - r^2) was tried first.  It
    // cancels: its error is ~1e-7 b^2 whatever the radius, so a zero-radius
    // sphere 12000 units away "was hit" by a ray passing 4.5 units from it, and
    // -- the hierarchy rightly never looking there -- frames differed between
    // scan forms (found by the mode's fuzz test).  The remedy term costs 6
    // more instructions per sphere and keeps the error ~1e-7 |dp_perp|^2.)
    //
    // |dp|^2 - r^2 decides on which side of the surface the origin lies, and a
    // ray leaving a radius-1000 sphere by 0.001 sits within one or two of its
    // rounding steps of zero: the reference's own self-intersection rule lives
    // in the last bits of this value, so it is rounded exactly as the
    // reference rounds it -- otherwise a few per cent of all paths through such
    // a scene take a different, if equally valid, course.
    //
    // The remedy term and the discriminant are rounded as the reference rounds
    // them too (round 3).  For a sphere many radii away, dp - d * b takes
    // numbers of the size of the distance down to the size of the radius, and
    // the discriminant's last bits -- which root distance comes out -- are
    // those of whichever sequence of roundings was used.  On configuration 5
    // (radius-0.3 spheres up to 60 units away, every bounce multiplying a
    // direction difference by ~100) the fused sequence ALONE put 0.09 % of the
    // frame's bytes more than one level from the exact frame; everything else
    // the mode changes together 0.014 %, a single fused multiply-add (the hit
    // point) 0.003 % (tools/fast_math_parity.py on builds with one group of
    // changes at a time, -DPT_X_DOT / _MADD / _SPHERE, -DPT_FAST_MINIMAL).
    // Sequential here: 0.044 % -- inside SURVEY 8(c)'s 99.9 % -- for 7 more
    // instructions per test (configuration 2: 48.2 -> 51.2 ms).  Switching by
    // distance (fused within 16 radii) was measured too: 0.077 % and 52.1 ms,
    // worse on both counts.  What stays cheap: v_sqrt_f32 and v_rcp_f32 as
    // they come, and the root ordering by min / max.
    const V3 dp = c - o;
    const float b = dot_seq(d, dp);
    const float cc = dot_seq(dp, dp) - r2;
    const V3 perp = dp - d * b;
    const float disc = r2 - dot_seq(perp, perp);
    if (disc < 0.0f)
        return false;
    const float s = __builtin_amdgcn_sqrtf(disc);
    const float q = (b >= 0.0f) ? (b + s) : (b - s);
    const float other = cc * __builtin_amdgcn_rcpf(q);
    const float t1 = __builtin_fminf(q, other);
    const float t2 = __builtin_fmaxf(q, other);
    dist = (t1 > 0.0f) ? t1 : t2;
    return !(t2 < 0.0f);
#else
    V3 dp = c - o;
    float ddp = dot_seq(d, dp);
    float dpp = dot_seq(dp, dp);
    dp = dp - d * ddp; // remedy term: component of dp orthogonal to d
    float dscr = r2 - dot_seq(dp, dp);
    if (dscr < 0.0f)
        return false;
#if defined(PT_X_SPHERE) && defined(__HIP_DEVICE_COMPILE__)
    dscr = sqrt_core(dscr >= 0x1.0p-96f ? dscr : 0x1.0p-96f); // (experiment: the exact root)
#else
    dscr = sqrt_rn(dscr);
#endif
    float q = (ddp >= 0.0f) ? (ddp + dscr) : (ddp - dscr);
    float a = q;
    float b = (dpp - r2) / q;
    if (a > b) {
        float keep = a;
        a = b;
        b = keep;
    }
    dist = (a > 0.0f) ? a : b;
    return !(a < 0.0f && b < 0.0f);
#endif
}

// glms_ray_triangle (Moeller-Trumbore, eps 1e-6, two-sided) with the edges
// e1 = v1 - v0 and e2 = v2 - v0 taken from the uploaded record.
// The same arithmetic in BOTH builds: the tolerance build does not cheapen the
// triangle test.  On a sliver (edges parallel to a part in 10^6) the
// determinant is rounding noise, and which noise decides hit or miss anywhere
// along the ray: the hierarchy's triangle boxes are sized for THIS sequence of
// roundings (csrc/bvh_build.hpp), and with fused cross products the linear
// scans "hit" slivers where the hierarchy had rightly not looked (found by the
// mode's fuzz test: frames differed between scan forms).
PT_HD bool ray_triangle(V3 o, V3 d, V3 v0, V3 e1, V3 e2, float &dist) {
    const float eps = 0.000001f;
    V3 p = cross_seq(d, e2);
    float det = dot_seq(e1, p);
    if (det > -eps && det < eps)
        return false;
    float inv_det = rcp_ieee(det);
    V3 t = o - v0;
    float u = inv_det * dot_seq(t, p);
    if (u < 0.0f || u > 1.0f)
        return false;
    V3 q = cross_seq(t, e1);
    float v = inv_det * dot_seq(d, q);
    if (v < 0.0f || u + v > 1.0f)
        return false;
    dist = inv_det * dot_seq(e2, q);
    return dist > eps;
}

struct Hit {
    float distance; // < 0: miss
    int object;     // index into the scene's object array, -1 on a miss
};

// Diagnostic build only (make EXTRA_HIPFLAGS=-DPT_FARREJECT_COUNT): how often
// could a WAVE skip the sqrt + divide half of a sphere test because every lane's
// root is already known to lie beyond its best hit?  The
// predicate here is the idealised one (no safety margins), so the count is an
// upper bound on what a conservative far-reject could skip.  tools/farreject_count.py.
#ifdef PT_FARREJECT_COUNT
__device__ unsigned long long g_farreject[4]; // wave tests with some dscr >= 0 | of those, all lanes far | lane tests | far lanes
#endif

// One object against the ray, then the accept rule of src/ray.c:25,45: keep a
// candidate unless (dist >= closest && closest > 0).
PT_HD void test_object(const SceneRecords &rec, V3 o, V3 d, uint32_t i,
                       const F4 g, Hit &h) {
    float dist = 0.0f;
    bool found;
    if (!is_triangle_record(g)) {
#if defined(PT_FARREJECT_COUNT) && defined(__HIP_DEVICE_COMPILE__) && !defined(PT_FAST_MATH)
        {
            const V3 dp = xyz(g) - o;
            const float ddp = dot(d, dp), dpp = dot(dp, dp);
            const V3 perp = dp - d * ddp;
            const float dscr = g.w - dot(perp, perp);
            const bool has = dscr >= 0.0f;
            const float best = h.distance, gap = ddp - best, cc = dpp - g.w;
            // chosen root >= best?  outside: ddp - s >= best; inside: ddp + s >= best;
            // behind (outside, ddp < 0): not found at all
            const bool far = has && best > 0.0f &&
                             (cc > 0.0f ? (ddp < 0.0f || (gap >= 0.0f && gap * gap >= dscr))
                                        : (gap >= 0.0f || dscr >= gap * gap));
            const unsigned long long has_mask = pt_ballot(has), far_mask = pt_ballot(far);
            if (has_mask != 0ull && (threadIdx.x & 63u) == (uint32_t)(__ffsll((long long)pt_ballot(true)) - 1)) {
                atomicAdd(&g_farreject[0], 1ull);
                if (far_mask == has_mask)
                    atomicAdd(&g_farreject[1], 1ull);
                atomicAdd(&g_farreject[2], (unsigned long long)__popcll(has_mask));
                atomicAdd(&g_farreject[3], (unsigned long long)__popcll(far_mask));
            }
        }
#endif
        found = ray_sphere(o, d, xyz(g), g.w, dist);
    } else {
        const uint32_t k = rec.link[i];
        const F4 v0 = rec.tri[4 * k + 0];
        const F4 e1 = rec.tri[4 * k + 1];
        const F4 e2 = rec.tri[4 * k + 2];
        found = ray_triangle(o, d, xyz(v0), xyz(e1), xyz(e2), dist);
    }
    // (one mask, two selects: as a branch the compiler selected twice, once by the
    // accept rule inside the branch and once more by `found` around it)
    const bool take = found && !(dist >= h.distance && h.distance > 0.0f);
    h.distance = take ? dist : h.distance;
    h.object = take ? (int)i : h.object;
}

// The scan of src/ray.c:12-65, every object tested in full.  Right for small
// scenes, where most objects are real candidates for most rays.
PT_HD Hit closest_hit(const SceneRecords &rec, V3 o, V3 d) {
    Hit h;
    h.distance = -1.0f;
    h.object = -1;
    // (Tried in round 2 and not kept: fetching four records per s_load_dwordx16
    // so that a wave waits for scalar memory once per four objects instead of
    // twice per object -- configuration 2 stayed at 62.6 ms: the other seven
    // waves of the SIMD cover those waits, the VALU issue slots are the limit.
    // Fetching record i + 1 during the test of record i cannot work at all: a
    // scalar load cannot be waited for individually.)
    for (uint32_t i = 0; i < rec.n_obj; ++i)
        test_object(rec, o, d, i, rec.geom[i], h);
    return h;
}

// The closest hit of a ray with a NaN or infinite component, which no box test
// can serve (src/ray.c's callers do produce such rays: a 0/0 root gives a NaN
// distance, the hit point d * NaN + o is NaN, and every later ray of the path
// starts there).  Working the reference's scan through for a ray with a NaN
// component: glms_ray_sphere's ddp = dot(d, c - o) is NaN, so is the remedy
// term, the discriminant and both roots; no comparison on the way rejects
// (NaN < 0, a > b and a < 0 && b < 0 are all false), so EVERY sphere is found,
// at distance NaN, and the accept rule keeps it (NaN >= closest is false).
// glms_ray_triangle ends in `dist > eps` with dist NaN (or bails out on a tiny
// finite determinant): never found.  The scan therefore returns the sphere
// with the highest index, whatever the scene looks like -- which one exact test
// reproduces, NaN bits included.  A scan of 64000 objects by one lane for each
// of the up to MAX_BOUNCES later rays of such a path used to cost a whole
// frame 150 ms.  Rays with an infinite but no NaN component (overflow) still
// get the sequential scan.
PT_HD Hit poisoned_hit(const SceneRecords &rec, V3 o, V3 d) {
    const bool has_nan = (o.x != o.x) | (o.y != o.y) | (o.z != o.z) |
                         (d.x != d.x) | (d.y != d.y) | (d.z != d.z);
    if (!has_nan)
        return closest_hit(rec, o, d);
    Hit h;
    h.distance = -1.0f;
    h.object = -1;
    if (rec.last_sphere != 0xFFFFFFFFu)
        test_object(rec, o, d, rec.last_sphere, rec.geom[rec.last_sphere], h);
    return h;
}

// glms_ray_sphere assumes a unit direction (src/ray.c:21 hands it whatever
// trace_ray was given); with any other length its roots are not where the ray
// meets the sphere, and no screen or box can say which objects it will "hit".
// The renderer only ever passes normalised directions (|d|^2 = 1 +- 3e-7) or,
// rarely, the zero vector glms_vec3_normalize returns for a tiny one; a caller
// of trace_ray may pass anything.  Such rays get the reference's own scan.
PT_HD bool direction_is_not_unit(V3 d) {
    const float len2 = d.x * d.x + d.y * d.y + d.z * d.z;
    return !(len2 >= 1.0f - 1.0e-5f && len2 <= 1.0f + 1.0e-5f); // (NaN: true)
}

// The same scan for large scenes, in two phases that keep the result (and the
// order objects are accepted in) identical:
//   screen   a wave-uniform pass over all objects, eight records fetched at a
//            time, with a CONSERVATIVE ray-sphere miss test: fused
//            multiply-adds, no remedy term, 14 VALU instructions instead of
//            21 + the 38 of sqrt/divide.  It may only say "certainly misses":
//            with err = 2^-16 * (|c-o|^2 + r^2) (>= 250x the rounding error of
//            either formulation, direction normalisation error included) a
//            sphere is dropped only if |dp_perp|^2 - r^2 > err.  NaN/overflow
//            compare false -> kept.  A triangle is screened through its
//            bounding sphere (radius inflated by 1e-3 at upload); a needle --
//            edges closer than 1e-2 rad, where the reference's test is noise
//            for every ray -- carries an infinite radius and is always kept
//            (csrc/ptgpu_abi.hip, bounding_sphere_record; the limit that
//            remains for rays lying in a triangle's plane: DESIGN.md 4.1).
//   resolve  each lane appends the indices of its surviving objects, in
//            ascending order, to its own queue in LDS (kScreenQueue entries);
//            when some lane's queue could overflow during the next block of
//            eight, and at the end, every lane runs the exact test_object()
//            on its own entries.  The divergent, expensive part thus runs on
//            queues that are dense across the wave instead of once per object
//            per wave.
constexpr uint32_t kScreenBlock = 8;
#ifndef PT_SCREEN_QUEUE
#define PT_SCREEN_QUEUE 12
#endif
constexpr uint32_t kScreenQueue = PT_SCREEN_QUEUE; // >= kScreenBlock + 1
constexpr uint32_t kLeafRef = 0x80000000u; // BVH refs: bit 31 = leaf (csrc/bvh_build.hpp)
#ifndef PT_LEAF_MAX
#define PT_LEAF_MAX 4
#endif
constexpr uint32_t kLeafSlots = PT_LEAF_MAX; // objects per leaf at most (= bvh_build.hpp's kLeafMax)

struct ScreenQueue {
    uint32_t *slot;   // this lane's entry 0 in LDS
    uint32_t stride;  // entries of one lane are `stride` words apart
    uint32_t *stack;  // BVH scan: this lane's traversal stack, same stride
};

#ifndef PT_BOX_SLACK
#define PT_BOX_SLACK 1.0e-4f
#endif
#ifndef PT_SCREEN_MARGIN
#define PT_SCREEN_MARGIN 0x1.0p-16f
#endif
__device__ __forceinline__ bool screen_keep(const F4 g, V3 o, V3 d) {
    // |w| is r^2 of the sphere itself, or of a triangle's bounding sphere:
    // one branch-free test for every record, so the eight fetches of a block
    // stay batched
    const float r2 = __builtin_fabsf(g.w);
    const float dx = g.x - o.x, dy = g.y - o.y, dz = g.z - o.z;
    const float ddp = __builtin_fmaf(dz, d.z, __builtin_fmaf(dy, d.y, dx * d.x));
    const float dpp = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
    const float perp2 = __builtin_fmaf(-ddp, ddp, dpp);
    // certainly misses iff (perp2 - r2) - 2^-16 * (dpp + r2) > 0
    const float excess = __builtin_fmaf(dpp + r2, -PT_SCREEN_MARGIN, perp2 - r2);
    return !(excess > 0.0f);
}

// One leaf of the hierarchy: its objects' records lie side by side in
// bvh_leaf_geom / bvh_order (padded), so all kLeafSlots of them are fetched at
// once -- one memory round trip, where reading order[] and then geom[order[]]
// object by object took up to eight in a row (half of the stream kernel's wave
// cycles were spent waiting for memory) -- screened, and the survivors of the
// first `count` appended to the lane's candidate queue.
__device__ __forceinline__ void visit_leaf(const SceneRecords &rec, uint32_t leaf_ref, V3 o, V3 d,
                                           uint32_t min_index, uint32_t *&wr, uint32_t stride) {
    const uint32_t count = (leaf_ref >> 28) & 7u;
    const uint32_t at = leaf_ref & 0x0FFFFFFFu;
    F4 g[kLeafSlots];
    uint32_t id[kLeafSlots];
#pragma unroll
    for (uint32_t k = 0; k < kLeafSlots; ++k) {
        g[k] = rec.bvh_leaf_geom[at + k];
        id[k] = rec.bvh_order[at + k];
    }
    // (every slot is screened, the padding ones too, and only then asked whether
    // it counts: written as one short-circuit condition the first record's fetch
    // was sunk into its branch, behind the wait for the indices -- a second
    // memory round trip per leaf visit)
    bool keep[kLeafSlots];
#pragma unroll
    for (uint32_t k = 0; k < kLeafSlots; ++k)
        keep[k] = screen_keep(g[k], o, d);
#pragma unroll
    for (uint32_t k = 0; k < kLeafSlots; ++k) {
        if (keep[k] && k < count && id[k] >= min_index) {
            *wr = id[k];
            wr += stride;
        }
    }
}

__device__ __forceinline__ Hit closest_hit_screened(const SceneRecords &rec,
                                                    V3 o, V3 d,
                                                    const ScreenQueue q) {
    Hit h;
    h.distance = -1.0f;
    h.object = -1;
    uint32_t *const first = q.slot;
    uint32_t *wr = first; // next free entry of this lane's queue

    auto resolve = [&]() {
        for (uint32_t *rd = first; pt_any(rd < wr); rd += q.stride) {
            if (rd < wr) {
                const uint32_t idx = *rd;
                test_object(rec, o, d, idx, load_record(rec.geom + idx), h);
            }
        }
        wr = first;
    };
    auto push = [&](bool keep, uint32_t i) {
        if (keep) {
            *wr = i;
            wr += q.stride;
        }
    };

    const uint32_t n = rec.n_obj;
    uint32_t *const nearly_full = first + (kScreenQueue - kScreenBlock) * q.stride;
    uint32_t i = 0;
    for (; i + kScreenBlock <= n; i += kScreenBlock) {
        F4 g[kScreenBlock];
#pragma unroll
        for (uint32_t j = 0; j < kScreenBlock; ++j)
            g[j] = rec.geom[i + j];
#pragma unroll
        for (uint32_t j = 0; j < kScreenBlock; ++j)
            push(screen_keep(g[j], o, d), i + j);
        if (pt_any(wr > nearly_full))
            resolve();
    }
    for (; i < n; ++i)
        push(screen_keep(rec.geom[i], o, d), i);
    resolve();
    // (the screen is geometry for a unit direction; any other gets the plain scan)
    const bool odd = direction_is_not_unit(d);
    if (pt_any(odd)) {
        if (odd)
            h = closest_hit(rec, o, d);
    }
    return h;
}

// ---------------------------------------------------------------------------
// BVH scan (not in the reference).
//
// The reference's accept rule walks the objects in index order:
//     keep candidate i unless (dist_i >= closest && closest > 0).
// Call a found candidate *positive* if dist > 0 and *degenerate* otherwise
// (dist is +0, -0 or NaN: the ray starts on the surface -- e.g. `dpp - r*r`
// rounds to exactly 0 next to a radius-1000 sphere, 0.3 % of configuration
// 5's rays -- or a root is 0/0).  Working the rule through:
//   * a degenerate candidate is always kept when met, and whatever found
//     candidate comes next is kept too (closest > 0 is false);
//   * between degenerates the rule is a running strict minimum.
// Hence, with L = the highest index of a degenerate candidate (none: L = -1),
// the walk returns the minimum-distance positive candidate among indices > L,
// lowest index on ties, and if there is none, candidate L itself.  That is a
// function of the candidate SET, so the hierarchy may deliver candidates in
// any order: a lane tracks the global best positive (distance, lowest index)
// and the highest degenerate index.  If the best positive lies beyond L it is
// the answer; if L is the last object, L is; otherwise (rare) the lane walks
// the hierarchy once more, considering only indices > L.
//
// Traversal: one inner node per lane per pass of the loop.  A node carries
// the boxes of both children, so one 64-byte fetch decides both; if the ray
// crosses both, the nearer one is entered and the other goes on the lane's
// stack in LDS (depth = tree depth, known at upload).  A box test is a slab
// test with slack proportional to the distances involved (the exact tests
// are themselves only accurate to ~|c-o| * 2^-23, so a far-away ray must be
// allowed to "hit" generously), written as "not certainly missed" so that a
// NaN can only widen the search; children wholly beyond the best exact
// distance so far are not entered.  Leaf objects go through the conservative
// sphere screen into the lane's LDS queue; exact tests run in dense resolve
// phases, as in the screened scan, early enough that `best` is known soon.
// Triangles can never be degenerate (their test demands distance > 1e-6).
// ---------------------------------------------------------------------------
#ifndef PT_BVH_RESOLVE_AT
#define PT_BVH_RESOLVE_AT 1u /* measured: 1 beats 0, 3 and 8 (C5 57 vs 58, 60, 62 ms) */
#endif

// One inner node: both children's boxes (binary16 on the device, rounded outward
// at upload) and refs, from two 16-byte records.
struct NodeBoxes {
    F4 l_lo, l_hi, r_lo, r_hi; // .w unused
    uint32_t l_ref, r_ref;
};
__device__ __forceinline__ float half_lo(uint32_t u) {
    return (float)__builtin_bit_cast(_Float16, (uint16_t)(u & 0xFFFFu));
}
__device__ __forceinline__ float half_hi(uint32_t u) {
    return (float)__builtin_bit_cast(_Float16, (uint16_t)(u >> 16));
}
__device__ __forceinline__ NodeBoxes load_node(const SceneRecords &rec, uint32_t node) {
    const F4 a = rec.bvh_nodes[2u * node + 0u];
    const F4 b = rec.bvh_nodes[2u * node + 1u];
    const uint32_t a0 = __builtin_bit_cast(uint32_t, a.x), a1 = __builtin_bit_cast(uint32_t, a.y),
                   a2 = __builtin_bit_cast(uint32_t, a.z);
    const uint32_t b0 = __builtin_bit_cast(uint32_t, b.x), b1 = __builtin_bit_cast(uint32_t, b.y),
                   b2 = __builtin_bit_cast(uint32_t, b.z);
    NodeBoxes n;
    n.l_lo = F4{half_lo(a0), half_hi(a0), half_lo(a1), 0.0f};
    n.l_hi = F4{half_hi(a1), half_lo(a2), half_hi(a2), 0.0f};
    n.r_lo = F4{half_lo(b0), half_hi(b0), half_lo(b1), 0.0f};
    n.r_hi = F4{half_hi(b1), half_lo(b2), half_hi(b2), 0.0f};
    n.l_ref = __builtin_bit_cast(uint32_t, a.w);
    n.r_ref = __builtin_bit_cast(uint32_t, b.w);
    return n;
}

// Slab test of one box for a ray given as 1/d (`inv`, approximate), -o/d
// (`oi`) and `slack0`: false only if the ray CERTAINLY misses the box or the box
// lies wholly beyond `best`.  t = lo * inv + oi is one fused multiply-add per
// plane where (lo - o) * inv is two operations; what the fused form loses is
// that lo - o is exact and lo * inv - o * inv is not: its error is up to one
// rounding of o * inv, 2^-24 |o / d| whatever the plane, so the ray carries
// slack0 = 2^-21 max |o / d| (eight such roundings) on top of the slack
// proportional to the distances involved.  Written as "not certainly missed":
// fminf / fmaxf drop a NaN operand (0 * inf on a box face, or o = 0 with
// d = 0 on an axis) and a NaN slack can only keep the box.
struct RaySlabs {
    V3 inv, oi;
    float slack0;
};
__device__ __forceinline__ RaySlabs ray_slabs(V3 o, V3 d) {
    RaySlabs r;
    r.inv = V3{__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y), __builtin_amdgcn_rcpf(d.z)};
    r.oi = V3{-o.x * r.inv.x, -o.y * r.inv.y, -o.z * r.inv.z};
    // A direction component that is +-0 (or too small for a finite reciprocal)
    // makes -o / d infinite, and lo * inf - inf is NaN on one plane of a slab
    // and -inf on the other where (lo - o) * inf was -inf and +inf: a box whose
    // slab holds the origin looked left behind (found by the differential run
    // against the linear scan: 17 of configuration 5's 447 million rays, 3
    // pixels).  Such an axis -- and one whose o / d overflows -- is dropped
    // instead: a NaN on both planes, which fminf / fmaxf ignore.  (An eye ray
    // with o = 0 gets there by itself: -0 * inf is NaN.)
    const float nan = __builtin_nanf("");
    r.oi.x = fabsf(r.oi.x) < __builtin_inff() ? r.oi.x : nan;
    r.oi.y = fabsf(r.oi.y) < __builtin_inff() ? r.oi.y : nan;
    r.oi.z = fabsf(r.oi.z) < __builtin_inff() ? r.oi.z : nan;
    r.slack0 = 0x1.0p-21f * fmaxf(fmaxf(fabsf(r.oi.x), fabsf(r.oi.y)), fabsf(r.oi.z));
    return r;
}
__device__ __forceinline__ bool box_crossed(const F4 lo, const F4 hi, V3 inv, V3 oi, float slack0,
                                            float best, float &enter) {
    const float x0 = __builtin_fmaf(lo.x, inv.x, oi.x), x1 = __builtin_fmaf(hi.x, inv.x, oi.x);
    const float y0 = __builtin_fmaf(lo.y, inv.y, oi.y), y1 = __builtin_fmaf(hi.y, inv.y, oi.y);
    const float z0 = __builtin_fmaf(lo.z, inv.z, oi.z), z1 = __builtin_fmaf(hi.z, inv.z, oi.z);
    enter = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fminf(z0, z1));
    const float leave = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fmaxf(z0, z1));
    const float slack = __builtin_fmaf(PT_BOX_SLACK, fabsf(enter) + fabsf(leave), slack0);
    // (leave - enter < -slack) | (leave < -slack), as one comparison.  An
    // infinite `enter` or `leave` (a product that overflows: box coordinates are
    // finite or point away from the box, and an infinite 1 / d was dropped above)
    // makes the slack infinite and every comparison false: the box is kept.
    const bool missed = (leave + slack < fmaxf(enter, 0.0f)) || (enter - slack > best);
    return !missed;
}

struct UnorderedBest {
    float best;     // smallest positive distance so far
    int best_idx;   // its object (lowest index among equals), -1: none
    float deg_dist; // distance (+-0 / NaN) of the degenerate candidate ...
    int deg_idx;    // ... with the highest index, -1: none
};

PT_HD void test_object_unordered(const SceneRecords &rec, V3 o, V3 d, uint32_t i,
                                 const F4 g, UnorderedBest &u) {
    float dist = 0.0f;
    bool found;
    if (!is_triangle_record(g)) {
        found = ray_sphere(o, d, xyz(g), g.w, dist);
    } else {
        const uint32_t k = rec.link[i];
        const F4 v0 = rec.tri[4 * k + 0];
        const F4 e1 = rec.tri[4 * k + 1];
        const F4 e2 = rec.tri[4 * k + 2];
        found = ray_triangle(o, d, xyz(v0), xyz(e1), xyz(e2), dist);
    }
    const bool positive = dist > 0.0f;
    const bool take_deg = found && !positive && (int)i > u.deg_idx;
    const bool take_best = found && positive && (dist < u.best || (dist == u.best && (int)i < u.best_idx));
    u.deg_idx = take_deg ? (int)i : u.deg_idx;
    u.deg_dist = take_deg ? dist : u.deg_dist;
    u.best = take_best ? dist : u.best;
    u.best_idx = take_best ? (int)i : u.best_idx;
}

// Diagnostic build only (make EXTRA_HIPFLAGS=-DPT_BVH_PROFILE): lane-passes and
// wave-passes of the three loops of the BVH scan, summed into device words no
// kernel reads (tools/bvh_profile.py prints them).  Not in the shipped library.
#ifdef PT_BVH_PROFILE
__device__ unsigned long long g_bvh_profile[12];
#define PT_BVH_COUNT(slot, active)                                             \
    do {                                                                       \
        prof[slot] += (uint32_t)__popcll(pt_ballot(active));                    \
        prof[(slot) + 1] += 1u;                                                \
    } while (0)
#else
#define PT_BVH_COUNT(slot, active)
#endif

__device__ __forceinline__ Hit closest_hit_bvh(const SceneRecords &rec, V3 o, V3 d,
                                               const ScreenQueue q) {
#ifdef PT_BVH_PROFILE
    uint32_t prof[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    prof[6] = (uint32_t)__popcll(pt_ballot(true));
    prof[7] = 1u;
#endif
    UnorderedBest u;
    u.best = __builtin_inff();
    u.best_idx = -1;
    u.deg_dist = 0.0f;
    u.deg_idx = -1;
    uint32_t min_index = 0; // only objects with index >= min_index count
    bool walking = true;    // this lane takes part in the current walk
    uint32_t *const first = q.slot;
    uint32_t *wr = first;
    // a leaf adds at most kLeafMax = 4 entries, so resolving once 8 of the 12
    // slots are used would be enough; resolving earlier (PT_BVH_RESOLVE_AT
    // entries) makes `best` known sooner, and nodes beyond it are skipped
    uint32_t *const nearly_full = first + (PT_BVH_RESOLVE_AT) * q.stride;

    auto resolve = [&]() {
        for (uint32_t *rd = first; pt_any(rd < wr); rd += q.stride) {
            PT_BVH_COUNT(4, rd < wr);
            if (rd < wr) {
                const uint32_t idx = *rd;
                test_object_unordered(rec, o, d, idx, load_record(rec.geom + idx), u);
            }
        }
        wr = first;
    };

    // A ray with a NaN or infinite component is not a geometric object any
    // more: the exact sphere test then "finds" spheres wherever its NaN
    // comparisons fall through (src/ray.c's callers do produce such rays: a
    // 0/0 root poisons the next ray of the path).  Boxes mean nothing for
    // it, so such a lane does the reference's sequential walk instead.
    // (the sum is non-finite iff some component is, or the sum overflows --
    // then the fallback is merely slower, still exact)  The same goes for a
    // direction that is not of unit length (direction_is_not_unit above).
    const float probe = (fabsf(o.x) + fabsf(o.y) + fabsf(o.z)) +
                        (fabsf(d.x) + fabsf(d.y) + fabsf(d.z));
    const bool poisoned = !(probe < __builtin_inff()) || direction_is_not_unit(d);
    walking = !poisoned;

    const RaySlabs slabs = ray_slabs(o, d);
    auto crosses = [&](const F4 lo, const F4 hi, float &enter) {
        return box_crossed(lo, hi, slabs.inv, slabs.oi, slabs.slack0, u.best, enter);
    };
    constexpr uint32_t kDone = 0xFFFFFFFFu; // not a valid ref: count field would be 7
    for (int walk = 0; walk < 2; ++walk) {
        if (!pt_any(walking))
            break;
        for (uint32_t k = 0; k < rec.n_unbounded; ++k) {
            const uint32_t idx = rec.bvh_unbounded[k];
            if (walking && idx >= min_index) {
                *wr = idx;
                wr += q.stride;
            }
            if (pt_any(wr > nearly_full))
                resolve();
        }
        uint32_t cur = walking ? rec.bvh_root : kDone;
        uint32_t *sp = q.stack; // next free stack entry of this lane
        while (pt_any(cur != kDone)) {
            // descend until no lane of the wave stands on an inner node: the
            // leaf code below then runs once for many lanes instead of on
            // every pass for a few
            while (pt_any(cur != kDone && !(cur & kLeafRef))) {
                PT_BVH_COUNT(0, cur != kDone && !(cur & kLeafRef));
                if (cur != kDone && !(cur & kLeafRef)) {
                    const NodeBoxes nb = load_node(rec, cur);
                    float l_enter, r_enter;
                    const bool go_l = crosses(nb.l_lo, nb.l_hi, l_enter);
                    const bool go_r = crosses(nb.r_lo, nb.r_hi, r_enter);
                    const uint32_t l_ref = nb.l_ref, r_ref = nb.r_ref;
                    // (decided on the comparisons' own masks before any branch: see bvh_walk_quantum)
                    const bool both = go_l && go_r;
                    const bool none = !go_l && !go_r;
                    const bool r_first = go_r && (!go_l || r_enter < l_enter);
                    if (both) {
                        *sp = r_first ? l_ref : r_ref; // the farther child waits
                        sp += q.stride;
                    }
                    cur = r_first ? r_ref : l_ref;
                    if (none) {
                        if (sp != q.stack) {
                            sp -= q.stride;
                            cur = *sp;
                        } else {
                            cur = kDone;
                        }
                    }
                }
            }
            PT_BVH_COUNT(2, cur != kDone);
            if (cur != kDone) { // a leaf
                visit_leaf(rec, cur, o, d, min_index, wr, q.stride);
                if (sp != q.stack) {
                    sp -= q.stride;
                    cur = *sp;
                } else {
                    cur = kDone;
                }
            }
            if (pt_any(wr > nearly_full))
                resolve();
        }
        resolve();
        // second walk only for lanes whose answer lies beyond a degenerate hit
        const bool again = walking && walk == 0 && u.deg_idx >= 0 &&
                           u.best_idx < u.deg_idx &&
                           (uint32_t)u.deg_idx + 1u < rec.n_obj;
        if (again) {
            min_index = (uint32_t)u.deg_idx + 1u;
            u.best = __builtin_inff();
            u.best_idx = -1;
        }
        walking = again;
    }

    Hit h;
    if (u.best_idx >= 0 && u.best_idx > u.deg_idx) {
        h.distance = u.best;
        h.object = u.best_idx;
    } else if (u.deg_idx >= 0) {
        h.distance = u.deg_dist;
        h.object = u.deg_idx;
    } else {
        h.distance = -1.0f;
        h.object = -1;
    }
    if (pt_any(poisoned)) {
        if (poisoned)
            h = poisoned_hit(rec, o, d);
    }
#ifdef PT_BVH_PROFILE
    if ((threadIdx.x & 63u) == 0u)
        for (int k = 0; k < 12; ++k)
            atomicAdd(&g_bvh_profile[k], (unsigned long long)prof[k]);
#endif
    return h;
}

// ---------------------------------------------------------------------------
// The same walk in resumable form, for the stream kernel (pt_kernels.hip).
//
// closest_hit_bvh() above runs a wave's 64 walks to completion together: the
// wave pays for its longest walk, and measured on configuration 5 the average
// lane takes part in only ~20 % of the inner-node steps its wave executes
// (tools/bvh_profile.py).  Here the walk's state lives in a struct, one call
// of bvh_walk_quantum() advances every walking lane by a bounded amount, and
// between calls the kernel hands finished lanes their next ray.  Same boxes,
// same screen, same exact tests, same accept rule: the hit is the same.
// ---------------------------------------------------------------------------
constexpr uint32_t kBvhDone = 0xFFFFFFFFu; // not a valid ref: count field would be 7

struct BvhWalk {
    V3 inv;             // 1 / direction (approximate: boxes only)
    V3 oi;              // -origin / direction ...
    float slack0;       // ... and the slack that pays for it (box_crossed)
    UnorderedBest u;
    uint32_t cur;       // ref being visited, kBvhDone: nothing left to visit
    uint32_t *sp;       // next free entry of this lane's stack
    uint32_t *wr;       // next free entry of this lane's candidate queue
    uint32_t min_index; // only objects with index >= min_index count (second walk)
};

__device__ __forceinline__ bool ray_is_poisoned(V3 o, V3 d) {
    const float probe = (fabsf(o.x) + fabsf(o.y) + fabsf(o.z)) +
                        (fabsf(d.x) + fabsf(d.y) + fabsf(d.z));
    return !(probe < __builtin_inff()) || direction_is_not_unit(d);
}

// Starts (or, with min_index > 0, restarts) a walk.  Objects outside the
// hierarchy are few or none.
__device__ __forceinline__ void bvh_walk_begin(const SceneRecords &rec,
                                               const ScreenQueue q, V3 o, V3 d,
                                               uint32_t min_index, BvhWalk &w) {
    w.min_index = min_index;
    w.cur = rec.bvh_root;
    w.sp = q.stack;
    w.wr = q.slot;
    // A few of them (configuration 5: the ground) go into the candidate queue and are
    // tested with the first leaf's candidates, in an exact-test stage that runs anyway
    // (C5: 31.0 -> 30.6 ms per pass); many are tested here and now -- the queue must
    // keep room for a leaf (kLeafSlots) below its "pressing" mark in both kernels.
    if (rec.n_unbounded <= 3u) {
        for (uint32_t k = 0; k < rec.n_unbounded; ++k) {
            const uint32_t idx = rec.bvh_unbounded[k];
            if (idx >= min_index) {
                *w.wr = idx;
                w.wr += q.stride;
            }
        }
        return;
    }
    for (uint32_t k = 0; k < rec.n_unbounded; ++k) {
        const uint32_t idx = rec.bvh_unbounded[k];
        if (idx >= min_index)
            test_object_unordered(rec, o, d, idx, load_record(rec.geom + idx), w.u);
    }
}

__device__ __forceinline__ void bvh_walk_first(const SceneRecords &rec,
                                               const ScreenQueue q, V3 o, V3 d,
                                               BvhWalk &w) {
    const RaySlabs slabs = ray_slabs(o, d);
    w.inv = slabs.inv;
    w.oi = slabs.oi;
    w.slack0 = slabs.slack0;
    w.u.best = __builtin_inff();
    w.u.best_idx = -1;
    w.u.deg_dist = 0.0f;
    w.u.deg_idx = -1;
    bvh_walk_begin(rec, q, o, d, 0u, w);
}

__device__ __forceinline__ bool bvh_walk_idle(const ScreenQueue q, const BvhWalk &w) {
    return w.cur == kBvhDone && w.wr == q.slot;
}

#ifdef PT_STAMPS
#define PT_QSTAMP(var)                                                         \
    unsigned long long var;                                                    \
    __builtin_amdgcn_sched_barrier(0);                                         \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory");  \
    __builtin_amdgcn_sched_barrier(0)
#define PT_QSTAMP_ADD(slot, later, earlier)                                    \
    do {                                                                       \
        if (prof != nullptr)                                                   \
            reinterpret_cast<unsigned long long *>(prof)[slot] += (later) - (earlier); \
    } while (0)
#else
#define PT_QSTAMP(var)
#define PT_QSTAMP_ADD(slot, later, earlier)
#endif

// One quantum for the lanes with `mine`: up to `inner_budget` inner-node
// steps (the loop also ends when no lane stands on an inner node), then one
// leaf for the lanes that stand on one, then exact tests of the queued
// candidates -- the latter two only when enough lanes take part.
__device__ __forceinline__ void bvh_walk_quantum(const SceneRecords &rec,
                                                 const ScreenQueue q, V3 o, V3 d,
                                                 bool mine, uint32_t inner_budget,
                                                 uint32_t leaf_lanes, uint32_t resolve_lanes,
                                                 BvhWalk &w, uint32_t *prof = nullptr) {
    (void)prof;
    // Leaves are visited once `leaf_lanes` (16) lanes stand on one (or no lane
    // can take an inner step), exact tests run once `resolve_lanes` (4) lanes
    // have candidates (or one has 8, or nothing else can move): measured on
    // configuration 5, worth 3 % over "as soon as any lane can".
    auto crosses = [&](const F4 lo, const F4 hi, float &enter) {
        return box_crossed(lo, hi, w.inv, w.oi, w.slack0, w.u.best, enter);
    };
    auto pop = [&]() {
        if (w.sp != q.stack) {
            w.sp -= q.stride;
            w.cur = *w.sp;
        } else {
            w.cur = kBvhDone;
        }
    };
    // (`mine` is implied wherever a walk has something to visit or to test: only a
    // walking lane's `cur` is not kBvhDone and only its queue is not empty -- and a
    // vote on one comparison is one instruction, where a vote on a conjunction
    // is written out as 0 / 1 and compared again)
    (void)mine;
    PT_QSTAMP(q_begin);
    for (uint32_t step = 0; step < inner_budget; ++step) {
        const bool inner = (int32_t)w.cur >= 0; // != kBvhDone and no leaf bit
        if (!pt_any(inner))
            break;
        PT_BVH_COUNT(0, inner);
        if (inner) {
            const NodeBoxes nb = load_node(rec, w.cur);
            float l_enter, r_enter;
            const bool go_l = crosses(nb.l_lo, nb.l_hi, l_enter);
            const bool go_r = crosses(nb.r_lo, nb.r_hi, r_enter);
            const uint32_t l_ref = nb.l_ref, r_ref = nb.r_ref;
            // (all the deciding done on the comparisons' own masks, before any
            // branch: kept across the three-way branch the two answers were
            // written out as 0 / 1 and compared again)
            const bool both = go_l && go_r;
            const bool none = !go_l && !go_r;
            const bool r_first = go_r && (!go_l || r_enter < l_enter);
            const uint32_t near = r_first ? r_ref : l_ref;
            if (both) {
                *w.sp = r_first ? l_ref : r_ref; // the farther child waits
                w.sp += q.stride;
            }
            w.cur = near;
            if (none)
                pop();
        }
    }
    PT_QSTAMP(q_inner);
    PT_QSTAMP_ADD(0, q_inner, q_begin);
    const bool at_leaf = w.cur != kBvhDone && (w.cur & kLeafRef);
    const unsigned long long leaf_mask = pt_ballot(at_leaf);
    const bool do_leaves = (uint32_t)__popcll(leaf_mask) >= leaf_lanes || !pt_any((int32_t)w.cur >= 0);
    if (do_leaves) {
        PT_BVH_COUNT(2, at_leaf);
    }
    if (do_leaves && at_leaf) {
        visit_leaf(rec, w.cur, o, d, w.min_index, w.wr, q.stride);
        pop();
    }
    PT_QSTAMP(q_leaf);
    PT_QSTAMP_ADD(1, q_leaf, q_inner);
    const bool pending = w.wr != q.slot;
    const bool pressing = w.wr > q.slot + (kScreenQueue - kLeafSlots - 1u) * q.stride; // + one more leaf still fits
    const bool can_step = w.cur != kBvhDone;
    if (pt_any(pressing) || (uint32_t)__popcll(pt_ballot(pending)) >= resolve_lanes ||
        (pt_any(pending) && !pt_any(can_step))) {
        for (uint32_t *rd = q.slot; pt_any(rd < w.wr); rd += q.stride) {
            PT_BVH_COUNT(4, rd < w.wr);
            if (rd < w.wr) {
                const uint32_t idx = *rd;
                test_object_unordered(rec, o, d, idx, load_record(rec.geom + idx), w.u);
            }
        }
        w.wr = q.slot;
    }
    PT_QSTAMP(q_exact);
    PT_QSTAMP_ADD(2, q_exact, q_leaf);
}

// For a lane whose walk is idle: either restarts it as the second walk (the
// answer lies beyond a degenerate candidate; returns false) or yields the hit.
__device__ __forceinline__ bool bvh_walk_finish(const SceneRecords &rec,
                                                const ScreenQueue q, V3 o, V3 d,
                                                BvhWalk &w, Hit &h) {
    const bool again = w.min_index == 0u && w.u.deg_idx >= 0 &&
                       w.u.best_idx < w.u.deg_idx &&
                       (uint32_t)w.u.deg_idx + 1u < rec.n_obj;
    if (again) {
        w.u.best = __builtin_inff();
        w.u.best_idx = -1;
        bvh_walk_begin(rec, q, o, d, (uint32_t)w.u.deg_idx + 1u, w);
        return false;
    }
    if (w.u.best_idx >= 0 && w.u.best_idx > w.u.deg_idx) {
        h.distance = w.u.best;
        h.object = w.u.best_idx;
    } else if (w.u.deg_idx >= 0) {
        h.distance = w.u.deg_dist;
        h.object = w.u.deg_idx;
    } else {
        h.distance = -1.0f;
        h.object = -1;
    }
    return true;
}

struct Surface {
    V3 point;
    V3 normal;
    uint32_t material;
};

// Point, geometric normal and material of an accepted hit (src/ray.c:28-36,
// 48-61): spheres get the outward normal normalize(point - centre);
// triangles the precomputed unit normal, turned to face the ray.
PT_HD Surface surface_at(const SceneRecords &rec, V3 o, V3 d, Hit h) {
    Surface s;
    const F4 g = load_record(rec.geom + h.object);
    const uint32_t meta = rec.meta[h.object];
    s.material = meta & kMetaMaterialMask;
#if defined(PT_FAST_MINIMAL) && PT_FAST_MINIMAL == 1
    s.point = V3{__builtin_fmaf(d.x, h.distance, o.x), __builtin_fmaf(d.y, h.distance, o.y),
                 __builtin_fmaf(d.z, h.distance, o.z)}; // the experiment's one fused operation
#elif defined(PT_FUSE)
    s.point = madd(d, h.distance, o);
#else
    s.point = d * h.distance + o;
#endif
    if (!is_triangle_record(g)) {
        s.normal = normalize(s.point - xyz(g));
    } else {
        const uint32_t k = rec.link[h.object];
        V3 n = xyz(rec.tri[4 * k + 3]);
        if (dot(n, d) > 0.0f)
            n = -n;
        s.normal = n;
    }
    return s;
}

} // namespace PT_NS
