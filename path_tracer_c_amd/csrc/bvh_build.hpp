// bvh_build.hpp -- host-side construction of the stackless bounding-volume
// hierarchy the BVH scan walks (csrc/device/pt_trace.h, closest_hit_bvh).
//
// Not in the reference: it scans its object array linearly (src/ray.c:12-65).
// SURVEY.md section 8(f)3 lists a BVH as the step after the linear scan is
// tuned.  The hierarchy only decides WHICH objects get the exact test; the
// exact tests and the accept rule are the reference's, so the image does not
// change.
//
// Layout: inner nodes only, 64 bytes each, each holding the boxes of its two
// children (so one fetch decides both, and the nearer child can be entered
// first):
//     L.lo.xyz, L.ref | L.hi.xyz, - | R.lo.xyz, R.ref | R.hi.xyz, -
// A ref is either an inner-node index (bit 31 clear) or a leaf:
// 0x80000000 | count << 28 | first, a run of `count` (1..kLeafMax) entries of
// `order[]`, which holds object indices.  `root` is the ref to start from
// (a leaf when the whole scene fits one), `depth` the deepest chain of inner
// nodes (= entries the traversal stack needs, at most).
// Boxes are those of the objects' bounding spheres, inflated (relative 1e-3
// plus an absolute term), so a ray the exact test can hit always crosses the
// box.  Objects without a finite box (NaN / infinite coordinates or radius)
// go to `unbounded[]` and are tested for every ray.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <vector>

#include "device/pt_layout.h"

namespace pt {

#ifndef PT_LEAF_MAX
#define PT_LEAF_MAX 4
#endif
constexpr uint32_t kLeafMax = PT_LEAF_MAX;

constexpr uint32_t kBvhLeafBit = 0x80000000u;

// ---- binary16 boxes for the device ------------------------------------------
// The walk is bound by the vector-memory path (64 bytes per clock per CU): a
// wave's inner-node step fetched 64 bytes per lane.  The device copy of a node
// therefore holds its two child boxes in binary16, rounded OUTWARD (lower
// corners toward -infinity, upper corners toward +infinity, beyond 65504 to
// +-infinity), which only ever enlarges a box -- by 2^-11 of a coordinate at
// most -- so the set of objects that reach the exact test can only grow, never
// lose one.  32 bytes per node, two 16-byte records:
//     {L.lo.x|L.lo.y, L.lo.z|L.hi.x, L.hi.y|L.hi.z, L.ref}  {R..., R.ref}
inline uint16_t half_magnitude(double a, bool away_from_zero) {
    if (!(a == a))
        return 0x7E00u;
    if (std::isinf(a))
        return 0x7C00u;
    if (a > 65504.0)
        return away_from_zero ? 0x7C00u : 0x7BFFu;
    if (a < 6.103515625e-05) { // below 2^-14: subnormal halves, steps of 2^-24
        const double n = a * 16777216.0;
        uint32_t k = (uint32_t)n;
        if (away_from_zero && n > (double)k)
            ++k;
        return (uint16_t)k; // 1024 = the smallest normal half: the encodings join up
    }
    int e = 0;
    const double m = std::frexp(a, &e) * 2.0; // a = m * 2^(e-1), m in [1, 2)
    e -= 1;
    const double q = (m - 1.0) * 1024.0;
    uint32_t k = (uint32_t)q;
    if (away_from_zero && q > (double)k)
        ++k;
    if (k == 1024u) {
        k = 0;
        ++e;
    }
    if (e > 15)
        return 0x7C00u;
    return (uint16_t)(((uint32_t)(e + 15) << 10) | k);
}

// x rounded to a binary16 value that is <= x (up = false) or >= x (up = true)
inline uint16_t half_directed(float x, bool up) {
    const bool negative = std::signbit(x);
    const uint16_t mag = half_magnitude(std::fabs((double)x), negative ? !up : up);
    return (uint16_t)(mag | (negative ? 0x8000u : 0u));
}

inline float half_value(uint16_t h) { // for the self-tests
    const uint32_t e = (h >> 10) & 31u, k = h & 1023u;
    double v;
    if (e == 0)
        v = std::ldexp((double)k, -24);
    else if (e == 31)
        v = k ? std::numeric_limits<double>::quiet_NaN() : std::numeric_limits<double>::infinity();
    else
        v = std::ldexp(1.0 + (double)k / 1024.0, (int)e - 15);
    return (float)((h & 0x8000u) ? -v : v);
}

// 4 float records per node (L.lo|ref, L.hi, R.lo|ref, R.hi) -> 2 packed records
inline std::vector<F4> pack_nodes_half(const std::vector<F4> &nodes) {
    auto bits = [](float f) {
        uint32_t u;
        std::memcpy(&u, &f, sizeof u);
        return u;
    };
    auto as_float = [](uint32_t u) {
        float f;
        std::memcpy(&f, &u, sizeof f);
        return f;
    };
    std::vector<F4> out(nodes.size() / 2);
    for (size_t n = 0; n + 3 < nodes.size(); n += 4) {
        for (int child = 0; child < 2; ++child) {
            const F4 &lo = nodes[n + 2 * child], &hi = nodes[n + 2 * child + 1];
            const uint32_t a = half_directed(lo.x, false) | ((uint32_t)half_directed(lo.y, false) << 16);
            const uint32_t b = half_directed(lo.z, false) | ((uint32_t)half_directed(hi.x, true) << 16);
            const uint32_t c = half_directed(hi.y, true) | ((uint32_t)half_directed(hi.z, true) << 16);
            out[n / 2 + child] = F4{as_float(a), as_float(b), as_float(c), as_float(bits(lo.w))};
        }
    }
    return out;
}

struct BvhHost {
    std::vector<F4> nodes;           // 4 records per inner node
    std::vector<uint32_t> order;     // object indices, leaf runs
    std::vector<uint32_t> unbounded; // objects tested for every ray
    uint32_t root = kBvhLeafBit;     // empty leaf: nothing to visit
    uint32_t depth = 0;
};

namespace bvh_detail {

struct Box {
    float lo[3], hi[3];
};

inline float as_float(uint32_t u) {
    float f;
    std::memcpy(&f, &u, sizeof f);
    return f;
}

struct Builder {
    const std::vector<Box> &boxes;
    std::vector<uint32_t> &items; // indices into boxes, permuted in place
    BvhHost &out;

    Box bounds(uint32_t begin, uint32_t end) const {
        Box all;
        for (int a = 0; a < 3; ++a) {
            all.lo[a] = std::numeric_limits<float>::infinity();
            all.hi[a] = -std::numeric_limits<float>::infinity();
        }
        for (uint32_t k = begin; k < end; ++k) {
            const Box &b = boxes[items[k]];
            for (int a = 0; a < 3; ++a) {
                all.lo[a] = std::min(all.lo[a], b.lo[a]);
                all.hi[a] = std::max(all.hi[a], b.hi[a]);
            }
        }
        return all;
    }

    static double half_area(const Box &b) {
        const double dx = (double)b.hi[0] - b.lo[0], dy = (double)b.hi[1] - b.lo[1],
                     dz = (double)b.hi[2] - b.lo[2];
        return dx * dy + dy * dz + dz * dx;
    }
    static void grow(Box &all, const Box &b) {
        for (int a = 0; a < 3; ++a) {
            all.lo[a] = std::min(all.lo[a], b.lo[a]);
            all.hi[a] = std::max(all.hi[a], b.hi[a]);
        }
    }
    static Box empty_box() {
        Box all;
        for (int a = 0; a < 3; ++a) {
            all.lo[a] = std::numeric_limits<float>::infinity();
            all.hi[a] = -std::numeric_limits<float>::infinity();
        }
        return all;
    }

    // Permutes items[begin, end) and returns the split position.  Surface-area
    // heuristic over 16 bins of the centres per axis: the split that minimises
    // area(left) * |left| + area(right) * |right|, so that a huge object (a
    // ground sphere of radius 1000 among unit spheres) is set apart near the
    // root instead of inflating every box above its leaf.  Below kSahLevels,
    // or when the centres cannot be separated, the median along the widest
    // axis (by index if all centres coincide), which bounds the depth.
    static constexpr uint32_t kSahLevels = 24;
#ifndef PT_BVH_BINS
#define PT_BVH_BINS 16
#endif
    static constexpr int kBins = PT_BVH_BINS;
    uint32_t choose_split(uint32_t begin, uint32_t end, uint32_t level) {
        const uint32_t n = end - begin;
        float clo[3], chi[3];
        for (int a = 0; a < 3; ++a) {
            clo[a] = std::numeric_limits<float>::infinity();
            chi[a] = -std::numeric_limits<float>::infinity();
        }
        auto centre = [&](uint32_t item, int a) {
            return 0.5f * boxes[item].lo[a] + 0.5f * boxes[item].hi[a];
        };
        for (uint32_t k = begin; k < end; ++k)
            for (int a = 0; a < 3; ++a) {
                const float c = centre(items[k], a);
                clo[a] = std::min(clo[a], c);
                chi[a] = std::max(chi[a], c);
            }
        if (level < kSahLevels) {
            double best_cost = std::numeric_limits<double>::infinity();
            int best_axis = -1, best_plane = 0;
            for (int a = 0; a < 3; ++a) {
                if (!(chi[a] > clo[a]))
                    continue;
                const double scale = (double)kBins / ((double)chi[a] - clo[a]);
                auto bin_of = [&](uint32_t item) {
                    int b = (int)(((double)centre(item, a) - clo[a]) * scale);
                    return b < 0 ? 0 : (b >= kBins ? kBins - 1 : b);
                };
                Box bin_box[kBins];
                uint32_t bin_n[kBins];
                for (int b = 0; b < kBins; ++b) {
                    bin_box[b] = empty_box();
                    bin_n[b] = 0;
                }
                for (uint32_t k = begin; k < end; ++k) {
                    const int b = bin_of(items[k]);
                    grow(bin_box[b], boxes[items[k]]);
                    ++bin_n[b];
                }
                double right_area[kBins];
                uint32_t right_n[kBins];
                Box acc = empty_box();
                uint32_t cnt = 0;
                for (int b = kBins - 1; b > 0; --b) {
                    if (bin_n[b]) {
                        grow(acc, bin_box[b]);
                        cnt += bin_n[b];
                    }
                    right_area[b] = cnt ? half_area(acc) : 0.0;
                    right_n[b] = cnt;
                }
                acc = empty_box();
                cnt = 0;
                for (int plane = 1; plane < kBins; ++plane) { // left = bins < plane
                    if (bin_n[plane - 1]) {
                        grow(acc, bin_box[plane - 1]);
                        cnt += bin_n[plane - 1];
                    }
                    if (cnt == 0 || right_n[plane] == 0)
                        continue;
                    const double cost = half_area(acc) * cnt + right_area[plane] * right_n[plane];
                    if (cost < best_cost) {
                        best_cost = cost;
                        best_axis = a;
                        best_plane = plane;
                    }
                }
            }
            if (best_axis >= 0 && std::isfinite(best_cost)) {
                const int a = best_axis;
                const double scale = (double)kBins / ((double)chi[a] - clo[a]);
                auto left = [&](uint32_t item) {
                    int b = (int)(((double)centre(item, a) - clo[a]) * scale);
                    b = b < 0 ? 0 : (b >= kBins ? kBins - 1 : b);
                    return b < best_plane;
                };
                const auto it = std::partition(items.begin() + begin, items.begin() + end, left);
                const uint32_t mid = (uint32_t)(it - items.begin());
                if (mid > begin && mid < end)
                    return mid;
            }
        }
        int axis = 0;
        for (int a = 1; a < 3; ++a)
            if (chi[a] - clo[a] > chi[axis] - clo[axis])
                axis = a;
        const uint32_t mid = begin + n / 2;
        if (chi[axis] > clo[axis])
            std::nth_element(items.begin() + begin, items.begin() + mid,
                             items.begin() + end, [&](uint32_t x, uint32_t y) {
                                 const float cx = boxes[x].lo[axis] + boxes[x].hi[axis];
                                 const float cy = boxes[y].lo[axis] + boxes[y].hi[axis];
                                 return cx < cy;
                             });
        return mid;
    }

    // emits the subtree over items[begin, end); returns its ref and box
    uint32_t build(uint32_t begin, uint32_t end, uint32_t level, Box &box) {
        box = bounds(begin, end);
        const uint32_t n = end - begin;
        if (n <= kLeafMax) {
            const uint32_t ref = kBvhLeafBit | (n << 28) | (uint32_t)out.order.size();
            for (uint32_t k = begin; k < end; ++k)
                out.order.push_back(items[k]);
            return ref;
        }
        out.depth = std::max(out.depth, level + 1);
        const uint32_t mid = choose_split(begin, end, level);
        const uint32_t me = (uint32_t)(out.nodes.size() / 4);
        out.nodes.resize(out.nodes.size() + 4);
        Box lbox, rbox;
        const uint32_t lref = build(begin, mid, level + 1, lbox);
        const uint32_t rref = build(mid, end, level + 1, rbox);
        out.nodes[4 * me + 0] = F4{lbox.lo[0], lbox.lo[1], lbox.lo[2], as_float(lref)};
        out.nodes[4 * me + 1] = F4{lbox.hi[0], lbox.hi[1], lbox.hi[2], 0.0f};
        out.nodes[4 * me + 2] = F4{rbox.lo[0], rbox.lo[1], rbox.lo[2], as_float(rref)};
        out.nodes[4 * me + 3] = F4{rbox.hi[0], rbox.hi[1], rbox.hi[2], 0.0f};
        return me;
    }
};

} // namespace bvh_detail

// `geom` = the uploaded geometry records: (centre, +r^2) for spheres,
// (bounding centre, -r^2) for triangles.  `corners` (optional): for object i
// with corners[6 i] <= corners[6 i + 3] its own axis-aligned extent
// {lo.xyz, hi.xyz} -- a triangle's vertices span much less than its bounding
// sphere -- which is then padded and intersected with the sphere's box.
// `apart` (optional, n_obj flags): objects kept out of the hierarchy and tested
// for every ray like the ones without a finite box -- the few spheres that
// dwarf the rest of a scene (a radius-1000 ground under unit-size objects), which
// nearly every ray meets anyway: in the tree such a sphere is a leaf at the root
// that every walk must visit, queue and test before it knows the distance that
// prunes everything else (round 3, csrc/ptgpu_abi.hip: spheres_that_dwarf_the_rest).
inline BvhHost build_bvh(const F4 *geom, uint32_t n_obj, const float *corners = nullptr,
                         const char *apart = nullptr) {
    using bvh_detail::Box;
    BvhHost out;
    std::vector<Box> boxes;
    std::vector<uint32_t> items;
    boxes.reserve(n_obj);
    std::vector<uint32_t> box_owner;
    for (uint32_t i = 0; i < n_obj; ++i) {
        const F4 &g = geom[i];
        const double r = std::sqrt((double)std::fabs(g.w));
        const double c[3] = {g.x, g.y, g.z};
        bool finite = std::isfinite(r);
        Box b;
        for (int a = 0; a < 3 && finite; ++a) {
            const double pad = r * 1.001 + 1e-6 * (std::fabs(c[a]) + r) + 1e-30;
            b.lo[a] = std::nextafter((float)(c[a] - pad), -std::numeric_limits<float>::infinity());
            b.hi[a] = std::nextafter((float)(c[a] + pad), std::numeric_limits<float>::infinity());
            finite = std::isfinite(c[a]) && std::isfinite(b.lo[a]) && std::isfinite(b.hi[a]);
        }
        // (experiment knob, DESIGN.md 4.1: what closing the in-plane limit would cost the
        // hierarchy -- PTGPU_TRI_BOX_RADII=k boxes a triangle by k bounding radii instead
        // of by its vertices' extent; unset = the shipped boxes)
        static const double tri_box_radii = [] {
            const char *e = std::getenv("PTGPU_TRI_BOX_RADII");
            return e != nullptr ? std::atof(e) : 0.0;
        }();
        if (finite && tri_box_radii > 0.0 && std::signbit(g.w)) {
            for (int a = 0; a < 3; ++a) {
                b.lo[a] = (float)(c[a] - tri_box_radii * r * 1.001);
                b.hi[a] = (float)(c[a] + tri_box_radii * r * 1.001);
            }
        } else
        if (finite && corners != nullptr && corners[6 * (size_t)i] <= corners[6 * (size_t)i + 3]) {
            // the exact test accepts points of the triangle only (up to its own
            // rounding, ~1e-6 of the coordinates): pad by 1e-3 of the largest
            // extent plus 1e-5 of the coordinates' magnitude
            const float *e = corners + 6 * (size_t)i;
            double span = 0.0, mag = 0.0;
            bool ok = true;
            for (int a = 0; a < 3; ++a) {
                ok = ok && std::isfinite(e[a]) && std::isfinite(e[a + 3]) && e[a] <= e[a + 3];
                span = std::max(span, (double)e[a + 3] - e[a]);
                mag = std::max(mag, std::max(std::fabs((double)e[a]), std::fabs((double)e[a + 3])));
            }
            if (ok) {
                const double pad = 1e-3 * span + 1e-5 * mag + 1e-30;
                for (int a = 0; a < 3; ++a) {
                    const float lo = std::nextafter((float)(e[a] - pad), -std::numeric_limits<float>::infinity());
                    const float hi = std::nextafter((float)(e[a + 3] + pad), std::numeric_limits<float>::infinity());
                    b.lo[a] = std::max(b.lo[a], lo); // never larger than the sphere's box
                    b.hi[a] = std::min(b.hi[a], hi);
                }
            }
        }
        if (!finite || (apart != nullptr && apart[i])) {
            out.unbounded.push_back(i);
            continue;
        }
        items.push_back((uint32_t)boxes.size());
        boxes.push_back(b);
        box_owner.push_back(i);
    }
    if (!items.empty()) {
        bvh_detail::Builder builder{boxes, items, out};
        Box all;
        out.root = builder.build(0, (uint32_t)items.size(), 0, all);
        for (uint32_t &o : out.order) // box index -> object index
            o = box_owner[o];
    }
    return out;
}

} // namespace pt
