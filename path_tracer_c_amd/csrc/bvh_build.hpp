// bvh_build.hpp -- host-side construction of the stackless bounding-volume
// hierarchy the BVH scan walks (csrc/device/pt_trace.h, closest_hit_bvh).
//
// Not in the reference: it scans its object array linearly (src/ray.c:12-65).
// SURVEY.md section 8(f)3 lists a BVH as the step after the linear scan is
// tuned.  The hierarchy only decides WHICH objects get the exact test; the
// exact tests and the accept rule are the reference's, so the image does not
// change.
//
// Layout: nodes in depth-first order, 32 bytes each,
//     lo.xyz, skip   |   hi.xyz, leaf
// `skip` = index of the first node after this node's subtree (where a ray
// that misses the box continues); the left child of an inner node is the next
// node.  `leaf` = 0 for an inner node, else (count << 24) | first, a run of
// `count` (1..kLeafMax) entries of `order[]`, which holds object indices.
// Boxes are those of the objects' bounding spheres, inflated (relative 1e-3
// plus an absolute term), so a ray the exact test can hit always crosses the
// box.  Objects without a finite box (NaN / infinite coordinates or radius)
// go to `unbounded[]` and are tested for every ray.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>

#include "device/pt_layout.h"

namespace pt {

constexpr uint32_t kLeafMax = 4;

struct BvhHost {
    std::vector<F4> nodes;           // 2 records per node
    std::vector<uint32_t> order;     // object indices, leaf runs
    std::vector<uint32_t> unbounded; // objects tested for every ray
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

    // emits the subtree over items[begin, end) and returns its node index
    uint32_t build(uint32_t begin, uint32_t end) {
        const uint32_t me = (uint32_t)(out.nodes.size() / 2);
        out.nodes.push_back(F4{0, 0, 0, 0});
        out.nodes.push_back(F4{0, 0, 0, 0});
        Box all, cent;
        for (int a = 0; a < 3; ++a) {
            all.lo[a] = cent.lo[a] = std::numeric_limits<float>::infinity();
            all.hi[a] = cent.hi[a] = -std::numeric_limits<float>::infinity();
        }
        for (uint32_t k = begin; k < end; ++k) {
            const Box &b = boxes[items[k]];
            for (int a = 0; a < 3; ++a) {
                all.lo[a] = std::min(all.lo[a], b.lo[a]);
                all.hi[a] = std::max(all.hi[a], b.hi[a]);
                const float c = 0.5f * b.lo[a] + 0.5f * b.hi[a];
                cent.lo[a] = std::min(cent.lo[a], c);
                cent.hi[a] = std::max(cent.hi[a], c);
            }
        }
        uint32_t leaf = 0;
        const uint32_t n = end - begin;
        int axis = 0;
        for (int a = 1; a < 3; ++a)
            if (cent.hi[a] - cent.lo[a] > cent.hi[axis] - cent.lo[axis])
                axis = a;
        const bool splittable = cent.hi[axis] > cent.lo[axis];
        if (n <= kLeafMax) {
            leaf = (n << 24) | (uint32_t)out.order.size();
            for (uint32_t k = begin; k < end; ++k)
                out.order.push_back(items[k]);
        } else {
            // (objects sharing one centre cannot be split spatially: they are
            // halved by index instead, so leaves never exceed kLeafMax)
            const uint32_t mid = begin + n / 2;
            if (splittable)
                std::nth_element(items.begin() + begin, items.begin() + mid,
                                 items.begin() + end, [&](uint32_t x, uint32_t y) {
                                     const float cx = boxes[x].lo[axis] + boxes[x].hi[axis];
                                     const float cy = boxes[y].lo[axis] + boxes[y].hi[axis];
                                     return cx < cy;
                                 });
            build(begin, mid); // left child = me + 1
            build(mid, end);
        }
        const uint32_t skip = (uint32_t)(out.nodes.size() / 2);
        out.nodes[2 * me + 0] = F4{all.lo[0], all.lo[1], all.lo[2], as_float(skip)};
        out.nodes[2 * me + 1] = F4{all.hi[0], all.hi[1], all.hi[2], as_float(leaf)};
        return me;
    }
};

} // namespace bvh_detail

// `geom` = the uploaded geometry records: (centre, +r^2) for spheres,
// (bounding centre, -r^2) for triangles.
inline BvhHost build_bvh(const F4 *geom, uint32_t n_obj) {
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
        if (!finite) {
            out.unbounded.push_back(i);
            continue;
        }
        items.push_back((uint32_t)boxes.size());
        boxes.push_back(b);
        box_owner.push_back(i);
    }
    if (!items.empty()) {
        bvh_detail::Builder builder{boxes, items, out};
        builder.build(0, (uint32_t)items.size());
        for (uint32_t &o : out.order) // box index -> object index
            o = box_owner[o];
    }
    return out;
}

} // namespace pt
