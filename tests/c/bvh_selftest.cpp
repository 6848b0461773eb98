// Host-only self-test of the BVH builder (csrc/bvh_build.hpp), built with g++
// under ASan + UBSan by tests/test_sanitizers.py.  Checks the structural
// invariants the device walk relies on, for benign and hostile inputs.
#include <cstdio>
#include <cstdlib>
#include <random>
#include <set>
#include <vector>

#include "bvh_build.hpp"

using pt::F4;

namespace {

int g_failures = 0;
#define CHECK(cond)                                                             \
    do {                                                                        \
        if (!(cond)) {                                                          \
            std::fprintf(stderr, "bvh_selftest: %s:%d: %s\n", __FILE__, __LINE__, #cond); \
            ++g_failures;                                                       \
        }                                                                       \
    } while (0)

uint32_t as_u32(float f) {
    uint32_t u;
    std::memcpy(&u, &f, sizeof u);
    return u;
}

struct Box {
    float lo[3], hi[3];
};

bool inside(const Box &inner, const Box &outer) {
    for (int a = 0; a < 3; ++a)
        if (!(inner.lo[a] >= outer.lo[a] && inner.hi[a] <= outer.hi[a]))
            return false;
    return true;
}

// Walks the subtree of `ref`: every object in it must lie inside `bound`
// (sphere box: centre +- radius); returns the chain length of inner nodes.
uint32_t walk(const pt::BvhHost &h, const std::vector<F4> &geom, uint32_t ref, const Box *bound,
              std::multiset<uint32_t> &seen, uint32_t level, std::vector<char> &visited) {
    if (ref & pt::kBvhLeafBit) {
        const uint32_t count = (ref >> 28) & 7u, first = ref & 0x0FFFFFFFu;
        CHECK(count >= 1 && count <= pt::kLeafMax);
        CHECK(first + count <= h.order.size());
        for (uint32_t k = 0; k < count && first + k < h.order.size(); ++k) {
            const uint32_t obj = h.order[first + k];
            CHECK(obj < geom.size());
            seen.insert(obj);
            if (bound != nullptr && obj < geom.size()) {
                const F4 &g = geom[obj];
                const float r = std::sqrt(std::fabs(g.w));
                Box b;
                const float c[3] = {g.x, g.y, g.z};
                for (int a = 0; a < 3; ++a) {
                    b.lo[a] = c[a] - r;
                    b.hi[a] = c[a] + r;
                }
                CHECK(inside(b, *bound));
            }
        }
        return 0;
    }
    CHECK(4u * ref + 3u < h.nodes.size());
    if (4u * ref + 3u >= h.nodes.size())
        return 0;
    CHECK(!visited[ref]); // a tree: every inner node has one parent
    visited[ref] = 1;
    CHECK(level < 4096u);
    const F4 *n = &h.nodes[4u * ref];
    Box l{{n[0].x, n[0].y, n[0].z}, {n[1].x, n[1].y, n[1].z}};
    Box r{{n[2].x, n[2].y, n[2].z}, {n[3].x, n[3].y, n[3].z}};
    if (bound != nullptr) { // children lie inside their parent's box
        CHECK(inside(l, *bound));
        CHECK(inside(r, *bound));
    }
    const uint32_t dl = walk(h, geom, as_u32(n[0].w), &l, seen, level + 1, visited);
    const uint32_t dr = walk(h, geom, as_u32(n[2].w), &r, seen, level + 1, visited);
    return 1u + (dl > dr ? dl : dr);
}

void check_tree(const std::vector<F4> &geom, const char *what) {
    const pt::BvhHost h = pt::build_bvh(geom.data(), (uint32_t)geom.size());
    std::multiset<uint32_t> seen;
    std::vector<char> visited(h.nodes.size() / 4u + 1u, 0);
    CHECK(h.nodes.size() % 4u == 0);
    uint32_t depth = 0;
    if (h.root != pt::kBvhLeafBit) // (empty leaf = nothing bounded)
        depth = walk(h, geom, h.root, nullptr, seen, 0, visited);
    CHECK(depth == h.depth);
    for (uint32_t u : h.unbounded) {
        CHECK(u < geom.size());
        seen.insert(u);
    }
    // every object exactly once, in a leaf or in the unbounded list
    CHECK(seen.size() == geom.size());
    for (uint32_t i = 0; i < geom.size(); ++i)
        CHECK(seen.count(i) == 1);
    // the depth bound the device stack is sized by: SAH levels + a balanced rest
    CHECK(h.depth <= 24u + 32u);
    if (g_failures)
        std::fprintf(stderr, "bvh_selftest: failed on: %s (%zu objects)\n", what, geom.size());
}

} // namespace

// The device copy of the nodes holds binary16 boxes rounded OUTWARD
// (pack_nodes_half): a lower corner may only move down, an upper corner only
// up, by one binary16 step at most, and the refs survive untouched.
void check_half_rounding() {
    std::mt19937 rng(99);
    std::uniform_real_distribution<float> mant(1.0f, 2.0f);
    std::vector<float> xs = {0.0f, -0.0f, 1.0f, -1.0f, 65504.0f, 65505.0f, -65519.9f, 65520.0f, 1e6f, -1e30f,
                             6.103515625e-05f, 6.1e-05f, 5.9604645e-08f, 3e-08f, -3e-08f, 1e-20f, 2047.9f,
                             std::numeric_limits<float>::infinity(), -std::numeric_limits<float>::infinity(),
                             std::numeric_limits<float>::denorm_min(), 0.1f, -0.1f, 1000.0f, -1001.8f};
    for (int e = -30; e <= 20; ++e)
        for (int k = 0; k < 200; ++k)
            xs.push_back(std::ldexp(mant(rng), e) * ((k & 1) ? -1.0f : 1.0f));
    for (float x : xs) {
        const float down = pt::half_value(pt::half_directed(x, false));
        const float up = pt::half_value(pt::half_directed(x, true));
        CHECK(down <= x && x <= up);
        if (std::fabs(x) <= 65504.0f) { // tight: one binary16 step (2^-10 relative, 2^-24 absolute)
            const double step = std::max(std::ldexp(std::fabs((double)x), -10), std::ldexp(1.0, -24));
            CHECK((double)x - down <= step && (double)up - x <= step);
            if (down == x || up == x)
                CHECK(down == up || (down < x) != (up > x) || down == up);
        }
    }
    CHECK(pt::half_value(pt::half_directed(1.0f, false)) == 1.0f);
    CHECK(pt::half_value(pt::half_directed(1.0f, true)) == 1.0f);
    CHECK(std::isinf(pt::half_value(pt::half_directed(70000.0f, true))));
    CHECK(pt::half_value(pt::half_directed(70000.0f, false)) == 65504.0f);
    CHECK(pt::half_value(pt::half_directed(-70000.0f, true)) == -65504.0f);
    // a packed tree: every packed child box contains the float one, refs equal
    std::uniform_real_distribution<float> pos(-2000.0f, 2000.0f), rad(0.05f, 30.0f);
    std::vector<F4> g;
    for (int i = 0; i < 5000; ++i) {
        const float r = rad(rng);
        g.push_back(F4{pos(rng), pos(rng), pos(rng), r * r});
    }
    g.push_back(F4{0.0f, -100000.0f, 0.0f, 1e10f}); // a box beyond the binary16 range
    const pt::BvhHost bvh = pt::build_bvh(g.data(), (uint32_t)g.size());
    const std::vector<F4> packed = pt::pack_nodes_half(bvh.nodes);
    CHECK(packed.size() * 2 == bvh.nodes.size());
    auto lo16 = [](float f) { return pt::half_value((uint16_t)(as_u32(f) & 0xFFFFu)); };
    auto hi16 = [](float f) { return pt::half_value((uint16_t)(as_u32(f) >> 16)); };
    for (size_t n = 0; n + 3 < bvh.nodes.size(); n += 4)
        for (int child = 0; child < 2; ++child) {
            const F4 &lo = bvh.nodes[n + 2 * child], &hi = bvh.nodes[n + 2 * child + 1];
            const F4 &p = packed[n / 2 + child];
            CHECK(as_u32(p.w) == as_u32(lo.w));
            CHECK(lo16(p.x) <= lo.x && hi16(p.x) <= lo.y && lo16(p.y) <= lo.z);
            CHECK(hi16(p.y) >= hi.x && lo16(p.z) >= hi.y && hi16(p.z) >= hi.z);
        }
}

int main() {
    check_half_rounding();
    std::mt19937 rng(12345);
    std::uniform_real_distribution<float> pos(-50.0f, 50.0f), rad(0.05f, 2.0f);
    const float inf = std::numeric_limits<float>::infinity();
    const float nan = std::numeric_limits<float>::quiet_NaN();

    check_tree({}, "empty scene");
    check_tree({F4{0, 0, 5, 1}}, "one sphere");
    for (uint32_t n : {2u, 3u, 4u, 5u, 9u, 64u, 65u, 1000u, 20000u}) {
        std::vector<F4> g;
        for (uint32_t i = 0; i < n; ++i) {
            const float r = rad(rng);
            g.push_back(F4{pos(rng), pos(rng), pos(rng), (i % 7 == 3 ? -1.0f : 1.0f) * r * r});
        }
        check_tree(g, "random spheres and triangle bounding spheres");
        g.push_back(F4{0, -1000, 0, 1000.0f * 1000.0f}); // a ground sphere among them
        check_tree(g, "with a huge sphere");
    }
    { // all centres equal: no spatial split exists
        std::vector<F4> g(300, F4{1, 2, 3, 0.25f});
        check_tree(g, "coincident objects");
    }
    { // centres on a line, growing radii: unbalanced SAH splits
        std::vector<F4> g;
        for (int i = 0; i < 3000; ++i)
            g.push_back(F4{(float)i * 1e-3f, 0, 0, (float)(i + 1) * (float)(i + 1)});
        check_tree(g, "nested spheres");
    }
    { // geometric progression of positions: every split peels one object off
        std::vector<F4> g;
        float x = 1.0f;
        for (int i = 0; i < 120; ++i, x *= 2.0f)
            g.push_back(F4{x, 0, 0, 1e-6f});
        check_tree(g, "geometric progression");
    }
    { // no finite box: NaN / infinite centre or radius, overflowing extent
        std::vector<F4> g;
        for (int i = 0; i < 100; ++i)
            g.push_back(F4{pos(rng), pos(rng), pos(rng), 1.0f});
        g.push_back(F4{inf, 0, 0, 1});
        g.push_back(F4{0, nan, 0, 1});
        g.push_back(F4{0, 0, 0, nan});
        g.push_back(F4{0, 0, 0, inf});
        g.push_back(F4{3e38f, 0, 0, 1e30f});
        g.push_back(F4{0, 0, 0, 0.0f});      // a point
        g.push_back(F4{1, 1, 1, -0.0f});     // a degenerate triangle record
        const pt::BvhHost h = pt::build_bvh(g.data(), (uint32_t)g.size());
        CHECK(h.unbounded.size() >= 4);
        check_tree(g, "objects without a finite box");
    }
    { // only unbounded objects
        std::vector<F4> g(5, F4{nan, 0, 0, 1});
        const pt::BvhHost h = pt::build_bvh(g.data(), (uint32_t)g.size());
        CHECK(h.root == pt::kBvhLeafBit && h.nodes.empty() && h.unbounded.size() == 5);
        check_tree(g, "nothing bounded");
    }
    if (g_failures) {
        std::fprintf(stderr, "bvh_selftest: %d failure(s)\n", g_failures);
        return 1;
    }
    std::printf("bvh_selftest: ok\n");
    return 0;
}
