// pt_duo.h -- pt_duokernel: the stream kernel with TWO paths per lane.
//
// (included by pt_kernels.hip, inside namespace PT_NS, after pt_streamkernel)
//
// pt_streamkernel (one path per lane) runs each stage of a walk -- inner-node
// steps, leaf visits, exact tests, shading -- with the lanes that happen to be
// in that stage: measured on configuration 5, 60 % / 41 % / 22 % / 47 % of a
// wave (profiles/r02, tools/bvh_profile.py).  Here a lane carries two
// (pixel, sample) paths, "slots" 0 and 1, each with its own walk state in
// registers and its own stack / candidate queue / bounce stack rows in LDS.
// Every stage serves, per lane, whichever of the lane's two slots is in that
// stage (slot 0 first): a lane idles in a stage only when NEITHER of its paths
// can take part, so with a stage's per-path probability p the lane use goes
// from p to about 1 - (1 - p)^2.  The stage code is the stream kernel's --
// same boxes, same screen, same exact tests, same accept rule, same shading --
// applied to a working copy picked from the two slots with v_cndmask and
// written back the same way; tasks, sample buffer and ordered sum are
// pt_samplekernel's.  Only which lane traces which path when differs, so the
// image is the same.
//
// A slot's states:
//   free    no path
//   fresh   owns a ray whose walk has not begun
//   walk    walking; a walk that has run out (nothing left to visit, no
//           candidate queued) is READY: the accept-rule decision, the shading
//           and the start of the next ray's walk happen in one stage.  A hit
//           that is known without a walk (primary-hit table, poisoned ray) is
//           written into the walk's result fields and is READY at once.
//   parked  path ended: `inv` holds the value its deepest call returns
#pragma once

namespace duo {
enum : uint32_t { kFree = 0, kFresh = 1, kWalk = 2, kParked = 3 };
constexpr uint32_t kKnownHit = 0xFFFFFFFFu; // min_index of a result that is not to be walked again
} // namespace duo

struct DuoSlot {
    V3 o, d;           // ray
    V3 inv;            // 1 / d while walking; once parked: the deepest call's value
    UnorderedBest u;
    uint32_t cur;      // ref being visited, kBvhDone: nothing left to visit
    uint32_t *sp, *wr; // next free stack entry / candidate-queue entry (LDS)
    uint32_t min_index;
    uint32_t depth, rng, st, out_index;
};

__device__ __forceinline__ float duo_pick(bool second, float a, float b) { return second ? b : a; }
__device__ __forceinline__ uint32_t duo_pick(bool second, uint32_t a, uint32_t b) { return second ? b : a; }
__device__ __forceinline__ int duo_pick(bool second, int a, int b) { return second ? b : a; }
template <class T> __device__ __forceinline__ T *duo_pick(bool second, T *a, T *b) { return second ? b : a; }
__device__ __forceinline__ V3 duo_pick(bool second, V3 a, V3 b) {
    return V3{second ? b.x : a.x, second ? b.y : a.y, second ? b.z : a.z};
}
__device__ __forceinline__ UnorderedBest duo_pick(bool second, const UnorderedBest &a, const UnorderedBest &b) {
    return UnorderedBest{second ? b.best : a.best, second ? b.best_idx : a.best_idx,
                         second ? b.deg_dist : a.deg_dist, second ? b.deg_idx : a.deg_idx};
}
template <class T> __device__ __forceinline__ void duo_put(bool c0, bool c1, T &a, T &b, T v) {
    a = c0 ? v : a;
    b = c1 ? v : b;
}
__device__ __forceinline__ void duo_put(bool c0, bool c1, V3 &a, V3 &b, V3 v) {
    duo_put(c0, c1, a.x, b.x, v.x);
    duo_put(c0, c1, a.y, b.y, v.y);
    duo_put(c0, c1, a.z, b.z, v.z);
}
__device__ __forceinline__ void duo_put(bool c0, bool c1, UnorderedBest &a, UnorderedBest &b,
                                        const UnorderedBest &v) {
    duo_put(c0, c1, a.best, b.best, v.best);
    duo_put(c0, c1, a.best_idx, b.best_idx, v.best_idx);
    duo_put(c0, c1, a.deg_dist, b.deg_dist, v.deg_dist);
    duo_put(c0, c1, a.deg_idx, b.deg_idx, v.deg_idx);
}

// A hit that needs no walk, written so that bvh_walk_finish() returns it.
__device__ __forceinline__ void duo_known_hit(Hit h, UnorderedBest &u, uint32_t &min_index) {
    u.best = h.distance;
    u.best_idx = h.object; // (-1: a miss; bvh_walk_finish then yields {-1, -1})
    u.deg_dist = 0.0f;
    u.deg_idx = -1;
    min_index = duo::kKnownHit;
}

#ifndef PT_DUO_WAVES
#define PT_DUO_WAVES 4
#endif
#ifndef PT_DUO_QUEUE
#define PT_DUO_QUEUE 8
#endif
constexpr uint32_t kDuoQueue = PT_DUO_QUEUE; // candidate-queue entries per slot (>= kLeafSlots + 2)

// Dynamic LDS of one 64-lane workgroup: the tally record, then per slot the
// bounce stack (uint16 per level), the candidate queue and the traversal stack.
__host__ __device__ inline uint32_t duo_slot_words(uint32_t max_bounces, uint32_t bvh_depth) {
    const uint32_t bounce16 = (2u * 64u * (max_bounces + 1u) + 15u) / 16u;
    return bounce16 * 4u + 64u * (kDuoQueue + bvh_depth + 2u);
}
__host__ __device__ inline size_t duo_lds_bytes(uint32_t max_bounces, uint32_t bvh_depth) {
    return 16u + (size_t)8 * duo_slot_words(max_bounces, bvh_depth);
}

__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(PT_DUO_WAVES, 8)))
    pt_duokernel(const LaunchArgs a, const F4 *__restrict__ scene_geom,
                 const F4 *__restrict__ scene_tri, const F4 *__restrict__ scene_mat,
                 const uint32_t *__restrict__ scene_meta, const uint32_t *__restrict__ scene_link,
                 float4 *__restrict__ out, uint32_t *__restrict__ chunk_counter) {
    extern __shared__ uint4 lds_dyn[];
    const uint32_t lane = threadIdx.x;
    constexpr uint32_t kStride = 64u;
    SceneRecords rec = records_in_global_memory(a.scene);
    rec.geom = scene_geom;
    rec.tri = scene_tri;
    rec.mat = scene_mat;
    rec.meta = scene_meta;
    rec.link = scene_link;
    const V3 sky = V3{a.scene.sky[0], a.scene.sky[1], a.scene.sky[2]};
    const uint32_t npix = a.num_rows * a.framew;

    // ---- LDS rows of the two slots ------------------------------------------
    uint4 *const tally = lds_dyn;
    uint32_t *const words = reinterpret_cast<uint32_t *>(lds_dyn + 1);
    const uint32_t slot_words = duo_slot_words(a.max_bounces, a.scene.bvh_depth);
    const uint32_t bounce_words = ((2u * 64u * (a.max_bounces + 1u) + 15u) / 16u) * 4u;
    uint16_t *const bounce0 = reinterpret_cast<uint16_t *>(words) + lane;
    uint16_t *const bounce1 = reinterpret_cast<uint16_t *>(words + slot_words) + lane;
    uint32_t *const queue0 = words + bounce_words + lane;
    uint32_t *const queue1 = queue0 + slot_words;
    uint32_t *const stack0 = queue0 + kDuoQueue * kStride;
    uint32_t *const stack1 = stack0 + slot_words;

    ChunkCursor cursor;
    DuoSlot s0, s1;
    s0.o = s0.d = s0.inv = splat(0.0f);
    s0.u = UnorderedBest{0.0f, -1, 0.0f, -1};
    s0.cur = kBvhDone;
    s0.sp = stack0;
    s0.wr = queue0;
    s0.min_index = 0u;
    s0.depth = 0u;
    s0.rng = 1u;
    s0.st = duo::kFree;
    s0.out_index = 0u;
    s1 = s0;
    s1.sp = stack1;
    s1.wr = queue1;
    uint32_t rays = 0, iters = 0, done = 0, shared = 0;
    Branches<false> no_branches;
#ifdef PT_BVH_PROFILE
    uint32_t prof[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif

    auto is_inner = [](uint32_t cur) { return cur != kBvhDone && !(cur & kLeafRef); };
    auto is_leaf = [](uint32_t cur) { return cur != kBvhDone && (cur & kLeafRef) != 0u; };
    // a slot whose candidate queue could not take another leaf's worth
    auto pressing0 = [&]() { return s0.wr > queue0 + (kDuoQueue - kLeafSlots - 1u) * kStride; };
    auto pressing1 = [&]() { return s1.wr > queue1 + (kDuoQueue - kLeafSlots - 1u) * kStride; };
    auto ready0 = [&]() { return s0.st == duo::kWalk && s0.cur == kBvhDone && s0.wr == queue0; };
    auto ready1 = [&]() { return s1.st == duo::kWalk && s1.cur == kBvhDone && s1.wr == queue1; };

    // Starts the walk of the ray (o, d) in the working copy (a poisoned ray
    // gets the reference's sequential scan at once, see closest_hit_bvh).
    auto begin_walk = [&](V3 o, V3 d, uint32_t *queue, uint32_t *stack, BvhWalk &w) {
        ++rays;
        if (ray_is_poisoned(o, d)) {
            duo_known_hit(poisoned_hit(rec, o, d), w.u, w.min_index);
            w.cur = kBvhDone;
            w.sp = stack;
            w.wr = queue;
        } else {
            bvh_walk_first(rec, ScreenQueue{queue, kStride, stack}, o, d, w);
        }
    };

    // ---- stage: finished walks -> accept rule, shading, next ray's walk begins ----
    auto shade_stage = [&]() {
        const bool c0 = ready0();
        const bool c1 = !c0 && ready1();
        if (!pt_any(c0 || c1))
            return;
        PT_BVH_COUNT(6, c0 || c1);
        V3 o = duo_pick(c1, s0.o, s1.o), d = duo_pick(c1, s0.d, s1.d);
        uint32_t *const queue = duo_pick(c1, queue0, queue1);
        uint32_t *const stack = duo_pick(c1, stack0, stack1);
        uint16_t *const bounce = duo_pick(c1, bounce0, bounce1);
        BvhWalk w;
        w.inv = duo_pick(c1, s0.inv, s1.inv);
        w.oi = splat(0.0f); // (this kernel's box test is the unfused one: (lo - o) * inv)
        w.slack0 = 0.0f;
        w.u = duo_pick(c1, s0.u, s1.u);
        w.cur = kBvhDone;
        w.sp = stack;
        w.wr = queue;
        w.min_index = duo_pick(c1, s0.min_index, s1.min_index);
        uint32_t depth = duo_pick(c1, s0.depth, s1.depth);
        uint32_t rng = duo_pick(c1, s0.rng, s1.rng);
        uint32_t st = duo::kWalk;
        if (c0 || c1) {
            const ScreenQueue q{queue, kStride, stack};
            Hit hit;
            if (bvh_walk_finish(rec, q, o, d, w, hit)) {
                PathState p;
                p.o = o;
                p.d = d;
                p.r = splat(0.0f);
                p.depth = depth;
                p.rng = rng;
                p.live = true;
                p.parked = false;
                advance_path<2, false>(rec, a, bounce, kStride, q, sky, p, no_branches, true, hit);
                o = p.o;
                d = p.d;
                depth = p.depth;
                rng = p.rng;
                if (p.live) {
                    begin_walk(o, d, queue, stack, w);
                } else {
                    st = duo::kParked;
                    w.inv = p.r;
                }
            }
#ifdef PT_BVH_PROFILE
            else
                atomicAdd(&g_bvh_profile[8], 1ull);
#endif
        }
        duo_put(c0, c1, s0.o, s1.o, o);
        duo_put(c0, c1, s0.d, s1.d, d);
        duo_put(c0, c1, s0.inv, s1.inv, w.inv);
        duo_put(c0, c1, s0.u, s1.u, w.u);
        duo_put(c0, c1, s0.cur, s1.cur, w.cur);
        duo_put(c0, c1, s0.sp, s1.sp, w.sp);
        duo_put(c0, c1, s0.wr, s1.wr, w.wr);
        duo_put(c0, c1, s0.min_index, s1.min_index, w.min_index);
        duo_put(c0, c1, s0.depth, s1.depth, depth);
        duo_put(c0, c1, s0.rng, s1.rng, rng);
        duo_put(c0, c1, s0.st, s1.st, st);
    };

    // ---- stage: ended paths unwind and store their sample ---------------------
    auto settle_stage = [&]() {
        const bool c0 = s0.st == duo::kParked;
        const bool c1 = !c0 && s1.st == duo::kParked;
        if (!pt_any(c0 || c1))
            return;
        if (c0 || c1) {
            PathState p;
            p.r = duo_pick(c1, s0.inv, s1.inv);
            p.depth = duo_pick(c1, s0.depth, s1.depth);
            const V3 sample = unwind_path(rec, duo_pick(c1, bounce0, bounce1), kStride, p);
            out[duo_pick(c1, s0.out_index, s1.out_index)] = make_float4(sample.x, sample.y, sample.z, 0.0f);
            ++done;
        }
        s0.st = c0 ? (uint32_t)duo::kFree : s0.st;
        s1.st = c1 ? (uint32_t)duo::kFree : s1.st;
    };

    // ---- stage: free slots fetch the next (pixel, sample) tasks ---------------
    auto refill_stage = [&]() {
        const bool c0 = s0.st == duo::kFree;
        const bool c1 = !c0 && s1.st == duo::kFree;
        bool is_free = c0 || c1;
        bool got = false;
        uint32_t sample = 0, j = 0;
        unsigned long long free_mask = pt_ballot(is_free);
        if (free_mask == 0ull || cursor.dry)
            return;
        while (free_mask != 0ull && !cursor.dry) {
            uint32_t smp, jj;
            if (take_task(cursor, a, chunk_counter, free_mask, is_free, lane, smp, jj) && jj < npix) {
                got = true;
                is_free = false;
                sample = smp;
                j = jj;
            }
            free_mask = pt_ballot(is_free);
        }
        if (!pt_any(got))
            return;
        const bool g0 = got && c0, g1 = got && c1;
        uint32_t *const queue = duo_pick(c1, queue0, queue1);
        uint32_t *const stack = duo_pick(c1, stack0, stack1);
        PathState p;
        p.o = p.d = p.r = splat(0.0f);
        p.depth = 0u;
        p.rng = 1u;
        uint32_t out_index = 0u;
        BvhWalk w;
        w.inv = w.oi = splat(0.0f);
        w.slack0 = 0.0f;
        w.u = UnorderedBest{0.0f, -1, 0.0f, -1};
        w.cur = kBvhDone;
        w.sp = stack;
        w.wr = queue;
        w.min_index = 0u;
        if (got) {
            start_path(a, j, sample, p, out_index);
            if (a.primary_hits != nullptr) { // the first segment's hit is known
                const float2 known = reinterpret_cast<const float2 *>(a.primary_hits)[j];
                duo_known_hit(Hit{known.x, __builtin_bit_cast(int, known.y)}, w.u, w.min_index);
                ++rays;
                ++shared;
            } else {
                begin_walk(p.o, p.d, queue, stack, w);
            }
        }
        duo_put(g0, g1, s0.o, s1.o, p.o);
        duo_put(g0, g1, s0.d, s1.d, p.d);
        duo_put(g0, g1, s0.inv, s1.inv, w.inv);
        duo_put(g0, g1, s0.u, s1.u, w.u);
        duo_put(g0, g1, s0.cur, s1.cur, w.cur);
        duo_put(g0, g1, s0.sp, s1.sp, w.sp);
        duo_put(g0, g1, s0.wr, s1.wr, w.wr);
        duo_put(g0, g1, s0.min_index, s1.min_index, w.min_index);
        duo_put(g0, g1, s0.depth, s1.depth, p.depth);
        duo_put(g0, g1, s0.rng, s1.rng, p.rng);
        duo_put(g0, g1, s0.out_index, s1.out_index, out_index);
        s0.st = g0 ? (uint32_t)duo::kWalk : s0.st;
        s1.st = g1 ? (uint32_t)duo::kWalk : s1.st;
    };

    // ---- one walk quantum: inner-node steps, a leaf visit, exact tests --------
    auto quantum = [&]() {
        // inner-node steps: up to walk_budget of them for each lane's chosen slot
        {
            const bool c0 = is_inner(s0.cur);
            const bool c1 = !c0 && is_inner(s1.cur);
            if (pt_any(c0 || c1)) {
                const V3 o = duo_pick(c1, s0.o, s1.o);
                const V3 inv = duo_pick(c1, s0.inv, s1.inv);
                const float best = duo_pick(c1, s0.u.best, s1.u.best);
                uint32_t cur = duo_pick(c1, s0.cur, s1.cur);
                uint32_t *sp = duo_pick(c1, s0.sp, s1.sp);
                uint32_t *const stack = duo_pick(c1, stack0, stack1);
                const bool mine = c0 || c1;
                auto crosses = [&](const F4 lo, const F4 hi, float &enter) {
                    const float x0 = (lo.x - o.x) * inv.x, x1 = (hi.x - o.x) * inv.x;
                    const float y0 = (lo.y - o.y) * inv.y, y1 = (hi.y - o.y) * inv.y;
                    const float z0 = (lo.z - o.z) * inv.z, z1 = (hi.z - o.z) * inv.z;
                    enter = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fminf(z0, z1));
                    const float leave = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fmaxf(z0, z1));
                    const float slack = PT_BOX_SLACK * (fabsf(enter) + fabsf(leave));
                    const bool missed = (enter == __builtin_inff()) | (leave == -__builtin_inff()) |
                                        (leave - enter < -slack) | (leave < -slack) |
                                        (enter - slack > best);
                    return !missed;
                };
                for (uint32_t step = 0; step < a.walk_budget; ++step) {
                    const bool inner = mine && is_inner(cur);
                    if (!pt_any(inner))
                        break;
                    PT_BVH_COUNT(0, inner);
                    if (inner) {
                        const NodeBoxes nb = load_node(rec, cur);
                        float l_enter, r_enter;
                        const bool go_l = crosses(nb.l_lo, nb.l_hi, l_enter);
                        const bool go_r = crosses(nb.r_lo, nb.r_hi, r_enter);
                        const uint32_t l_ref = nb.l_ref, r_ref = nb.r_ref;
                        if (go_l & go_r) {
                            const bool r_first = r_enter < l_enter;
                            *sp = r_first ? l_ref : r_ref; // the farther child waits
                            sp += kStride;
                            cur = r_first ? r_ref : l_ref;
                        } else if (go_l | go_r) {
                            cur = go_l ? l_ref : r_ref;
                        } else if (sp != stack) {
                            sp -= kStride;
                            cur = *sp;
                        } else {
                            cur = kBvhDone;
                        }
                    }
                }
                duo_put(c0, c1, s0.cur, s1.cur, cur);
                duo_put(c0, c1, s0.sp, s1.sp, sp);
            }
        }
        // one leaf for the lanes that stand on one (a slot whose queue is
        // nearly full waits for its exact tests first)
        {
            const bool l0 = is_leaf(s0.cur) && !pressing0();
            const bool l1 = is_leaf(s1.cur) && !pressing1();
            const bool c0 = l0, c1 = !l0 && l1;
            const unsigned long long leaf_mask = pt_ballot(c0 || c1);
            const bool do_leaves = (uint32_t)__popcll(leaf_mask) >= a.leaf_lanes ||
                                   !pt_any(is_inner(s0.cur) || is_inner(s1.cur));
            if (do_leaves && leaf_mask != 0ull) {
                PT_BVH_COUNT(2, c0 || c1);
                uint32_t cur = duo_pick(c1, s0.cur, s1.cur);
                uint32_t *sp = duo_pick(c1, s0.sp, s1.sp);
                uint32_t *wr = duo_pick(c1, s0.wr, s1.wr);
                if (c0 || c1) {
                    visit_leaf(rec, cur, duo_pick(c1, s0.o, s1.o), duo_pick(c1, s0.d, s1.d),
                               duo_pick(c1, s0.min_index, s1.min_index), wr, kStride);
                    if (sp != duo_pick(c1, stack0, stack1)) {
                        sp -= kStride;
                        cur = *sp;
                    } else {
                        cur = kBvhDone;
                    }
                }
                duo_put(c0, c1, s0.cur, s1.cur, cur);
                duo_put(c0, c1, s0.sp, s1.sp, sp);
                duo_put(c0, c1, s0.wr, s1.wr, wr);
            }
        }
        // exact tests of the queued candidates
        {
            const bool p0 = s0.wr != queue0, p1 = s1.wr != queue1;
            const bool press0 = pressing0(), press1 = pressing1();
            const bool can_step = is_inner(s0.cur) || is_inner(s1.cur) ||
                                  (is_leaf(s0.cur) && !press0) || (is_leaf(s1.cur) && !press1);
            if (pt_any(press0 || press1) ||
                (uint32_t)__popcll(pt_ballot(p0 || p1)) >= a.resolve_lanes ||
                (pt_any(p0 || p1) && !pt_any(can_step))) {
                const bool c1 = press1 || (!p0 && p1);
                const bool c0 = !c1 && p0;
                const V3 o = duo_pick(c1, s0.o, s1.o), d = duo_pick(c1, s0.d, s1.d);
                UnorderedBest u = duo_pick(c1, s0.u, s1.u);
                uint32_t *const queue = duo_pick(c1, queue0, queue1);
                uint32_t *const wr = duo_pick(c1, s0.wr, s1.wr);
                const bool mine = c0 || c1;
                for (uint32_t *rd = queue; pt_any(mine && rd < wr); rd += kStride) {
                    PT_BVH_COUNT(4, mine && rd < wr);
                    if (mine && rd < wr) {
                        const uint32_t idx = *rd;
                        test_object_unordered(rec, o, d, idx, rec.geom[idx], u);
                    }
                }
                duo_put(c0, c1, s0.u, s1.u, u);
                s0.wr = c0 ? queue0 : s0.wr;
                s1.wr = c1 ? queue1 : s1.wr;
            }
        }
    };

    for (;;) {
        const bool r0 = ready0(), r1 = ready1();
        const bool walking0 = s0.st == duo::kWalk && !r0, walking1 = s1.st == duo::kWalk && !r1;
        const bool idle_free = !cursor.dry;
        const bool wait0 = r0 || s0.st == duo::kParked || (s0.st == duo::kFree && idle_free);
        const bool wait1 = r1 || s1.st == duo::kParked || (s1.st == duo::kFree && idle_free);
        const unsigned long long walking = pt_ballot(walking0 || walking1);
        const bool full_service =
            walking == 0ull || (uint32_t)__popcll(pt_ballot(wait0 || wait1)) >= a.settle_lanes;
        if (!full_service && (uint32_t)__popcll(pt_ballot(r0 || r1)) >= a.shade_lanes) {
            shade_stage();
            continue;
        }
        if (full_service) {
#pragma nounroll
            for (int round = 0; round < 2; ++round) {
                shade_stage();
                settle_stage();
                refill_stage();
                // a second round while enough slots still wait: a lane serves
                // one slot per stage
                const bool again0 = ready0() || s0.st == duo::kParked || (s0.st == duo::kFree && !cursor.dry);
                const bool again1 = ready1() || s1.st == duo::kParked || (s1.st == duo::kFree && !cursor.dry);
                if ((uint32_t)__popcll(pt_ballot(again0 || again1)) < a.reshade_lanes)
                    break;
            }
            if (pt_ballot(s0.st != duo::kFree || s1.st != duo::kFree) == 0ull)
                break; // pool dry and every slot idle
            continue;
        }
        ++iters;
        quantum();
    }
#ifdef PT_BVH_PROFILE
    if (lane == 0u)
        for (int k = 0; k < 12; ++k)
            atomicAdd(&g_bvh_profile[k], (unsigned long long)prof[k]);
#endif
    count_block(a.counters, tally, rays, rays - shared, done, iters, lane, 64u);
}
