// ptgpu_abi.hip -- the C-ABI of the MI355X render path (include/ptgpu.h) and
// the legacy entry points it stands behind (include/path_tracer_api.h:
// render_task, trace_ray; threadpool_* live in host/threadpool.c and call
// ptgpu_pool_render_rows here).
//
// What replaces what (reference file:line):
//   threadpool_add_task / _wait_for_tasks   src/threadpool.c:84-101
//     -> host/threadpool.c: rows are queued on the host; wait_for_tasks renders all queued rows
//        of a frame with one megakernel launch per contiguous row run and
//        copies them into the caller's frame.  It returns when every queued
//        task has completed (the intended contract; the reference returns
//        early and can lose the wake-up).
//   render_task                              src/renderer.c:92-116
//     -> a one-row launch of the same kernel.
//   trace_ray                                src/ray.c:4-73
//     -> a one-ray launch of the batched query kernel.
// There is no CPU implementation of any of these in this library: if HIP is
// unusable the explicit API returns PTGPU_ERR_NO_DEVICE and the void legacy
// functions print the reason and abort().
#include <hip/hip_runtime.h>
#include <rccl/rccl.h> // types and prototypes only: librccl is dlopen()ed on first use

#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <limits>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "bvh_build.hpp"
#include "device/pt_launch.h"
#include "device/pt_vec.h"
#include "host/pool_rows.h"
#include "ptgpu.h"

namespace {

thread_local std::string g_last_error = "";

int fail(int code, const std::string &what) {
    g_last_error = what;
    return code;
}

int fail_hip(hipError_t e, const char *where) {
    std::string msg = std::string(where) + ": " + hipGetErrorString(e);
    const bool no_device = (e == hipErrorNoDevice || e == hipErrorInvalidDevice ||
                            e == hipErrorInsufficientDriver ||
                            e == hipErrorNotInitialized);
    return fail(no_device ? PTGPU_ERR_NO_DEVICE : PTGPU_ERR_HIP, msg);
}

#define PT_TRY_HIP(expr)                                                       \
    do {                                                                       \
        hipError_t pt_e_ = (expr);                                             \
        if (pt_e_ != hipSuccess)                                               \
            return fail_hip(pt_e_, #expr);                                     \
    } while (0)

constexpr int kMaxDevices = 64;
constexpr size_t kLdsBudget = 160u * 1024u; // one workgroup may own it all
constexpr uint32_t kBvhMinObjects = 64;      // a hierarchy is built from here up
constexpr uint32_t kBvhAutoObjects = 96;    // and used by default from here up
                                             // (measured: slower than the screened
                                             // linear scan at 250 objects, 1.2x
                                             // faster at 1000, 5.9x at 16000)

// Per-device statistics block, allocated on first use.
std::mutex g_ctx_mutex;
pt::Counters *g_counters[kMaxDevices] = {nullptr};

int current_device(int *dev) {
    PT_TRY_HIP(hipGetDevice(dev));
    if (*dev < 0 || *dev >= kMaxDevices)
        return fail(PTGPU_ERR_LIMIT, "device ordinal out of range");
    return PTGPU_OK;
}

int counters_for_current_device(pt::Counters **out) {
    int dev = 0;
    int rc = current_device(&dev);
    if (rc != PTGPU_OK)
        return rc;
    std::lock_guard<std::mutex> guard(g_ctx_mutex);
    if (g_counters[dev] == nullptr) {
        pt::Counters *c = nullptr;
        PT_TRY_HIP(hipMalloc(&c, sizeof(pt::Counters) * pt::kCounterShards));
        PT_TRY_HIP(hipMemset(c, 0, sizeof(pt::Counters) * pt::kCounterShards));
        g_counters[dev] = c;
    }
    *out = g_counters[dev];
    return PTGPU_OK;
}

// Sum of the counter shards (synchronous copy on the default stream).
int read_counters(const pt::Counters *device, pt::Counters *total) {
    std::vector<pt::Counters> shards(pt::kCounterShards);
    PT_TRY_HIP(hipMemcpy(shards.data(), device, sizeof(pt::Counters) * pt::kCounterShards,
                         hipMemcpyDeviceToHost));
    std::memset(total, 0, sizeof *total);
    for (const pt::Counters &c : shards) {
        total->rays += c.rays;
        total->samples += c.samples;
        total->wave_iters += c.wave_iters;
        total->scans += c.scans;
    }
    return PTGPU_OK;
}

pt::V3 to_v3(const vec3s &v) { return pt::V3{v.x, v.y, v.z}; }

// geom record of a triangle: its smallest enclosing sphere (midpoint of the
// longest edge if the triangle is right or obtuse, else the circumcentre),
// computed in double, the radius inflated by 1e-3 and r^2 rounded up, stored
// as (centre, -r^2).  Only the conservative screen of the large-scene scan
// uses it; a degenerate or non-finite triangle gets an infinite radius
// (always kept), so screening can never drop what the exact test would hit.
// A triangle the screen and the hierarchy must never cull: (0, 0, 0, -inf) keeps
// every ray in screen_keep and has no finite box (bvh_build.hpp: `unbounded`).
pt::F4 never_culled_triangle_record() {
    return pt::F4{0.0f, 0.0f, 0.0f, -std::numeric_limits<float>::infinity()};
}

// sin^2 of the angle the two uploaded edges enclose (0 for a zero edge, NaN for NaN)
double triangle_sin2(const triangle &t) {
    const float e1f[3] = {t.v1.x - t.v0.x, t.v1.y - t.v0.y, t.v1.z - t.v0.z}; // as uploaded
    const float e2f[3] = {t.v2.x - t.v0.x, t.v2.y - t.v0.y, t.v2.z - t.v0.z};
    const double a[3] = {e1f[0], e1f[1], e1f[2]}, b[3] = {e2f[0], e2f[1], e2f[2]};
    const double n[3] = {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2],
                         a[0] * b[1] - a[1] * b[0]};
    const double n2 = n[0] * n[0] + n[1] * n[1] + n[2] * n[2];
    const double a2 = a[0] * a[0] + a[1] * a[1] + a[2] * a[2];
    const double b2 = b[0] * b[0] + b[1] * b[1] + b[2] * b[2];
    if (!(a2 * b2 > 0.0))
        return a2 * b2 == 0.0 ? 0.0 : std::numeric_limits<double>::quiet_NaN();
    return n2 / (a2 * b2);
}

constexpr double kNeedleSin2 = 1e-4; // edges closer than 1e-2 rad: never culled

pt::F4 bounding_sphere_record(const triangle &t) {
    const double p[3][3] = {{t.v0.x, t.v0.y, t.v0.z},
                            {t.v1.x, t.v1.y, t.v1.z},
                            {t.v2.x, t.v2.y, t.v2.z}};
    // Needles.  The reference's triangle test (glms_ray_triangle, src/ray.c:42)
    // divides by det = e1 . (d x e2), whose rounding error is a few 1e-7 of
    // |e1| |e2| whatever the ray.  When the edges are parallel to within that
    // (a sliver: v1 ~ v2, or a zero-area triangle) det is noise for EVERY ray,
    // u, v and the distance are noise too, and the test "hits" the triangle
    // wherever a ray comes close to the LINE through it, however far from the
    // three vertices -- found by the round-2 fuzz: a sliver of 67 units with
    // v1 and v2 6e-5 apart, hit 90 units beyond its end.  No finite bound
    // holds such hits, so a triangle whose edges enclose an angle below 1e-2
    // is tested for every ray, as the reference tests everything.  Wider
    // triangles keep their bounds.  There the same noise -- about
    // 6e-8 * (distance / edge) / (sin(angle of the edges) * |cos(ray, normal)|)
    // in the barycentric coordinates -- needs a ray that lies in the triangle's
    // plane to ~2e-4 rad, from tens of edge lengths away, AND passes the
    // triangle within a fraction of an edge: tools/needle_probe.py aims rays
    // like that, and 2 of 6 million differ from the plain scan (edges 0.05 rad
    // apart; none at 0.1 and wider); no random scene has produced one.  The
    // plain scan culls nothing and is exact whatever the ray.
    if (!(triangle_sin2(t) >= kNeedleSin2)) // sin^2 < 1e-4, zero edge, NaN
        return never_culled_triangle_record();
    auto dist2 = [](const double *a, const double *b) {
        double s = 0;
        for (int i = 0; i < 3; ++i)
            s += (a[i] - b[i]) * (a[i] - b[i]);
        return s;
    };
    double c[3] = {0, 0, 0};
    // longest edge and the vertex opposite to it
    int ia = 0, ib = 1, ic = 2;
    double longest = dist2(p[0], p[1]);
    if (dist2(p[1], p[2]) > longest) { longest = dist2(p[1], p[2]); ia = 1; ib = 2; ic = 0; }
    if (dist2(p[2], p[0]) > longest) { longest = dist2(p[2], p[0]); ia = 2; ib = 0; ic = 1; }
    for (int i = 0; i < 3; ++i)
        c[i] = 0.5 * (p[ia][i] + p[ib][i]);
    double r2 = 0.25 * longest;
    if (dist2(c, p[ic]) > r2) { // acute: circumcentre
        double ab[3], ac[3], n[3];
        for (int i = 0; i < 3; ++i) { ab[i] = p[1][i] - p[0][i]; ac[i] = p[2][i] - p[0][i]; }
        n[0] = ab[1] * ac[2] - ab[2] * ac[1];
        n[1] = ab[2] * ac[0] - ab[0] * ac[2];
        n[2] = ab[0] * ac[1] - ab[1] * ac[0];
        const double n2 = n[0] * n[0] + n[1] * n[1] + n[2] * n[2];
        const double ab2 = ab[0] * ab[0] + ab[1] * ab[1] + ab[2] * ab[2];
        const double ac2 = ac[0] * ac[0] + ac[1] * ac[1] + ac[2] * ac[2];
        // centre - p0 = (|ac|^2 (n x ab) ... ) standard form:
        // ((ab2 * ac) - (ac2 * ab)) x n / (2 n2), with sign arranged below
        double u[3] = {ab2 * ac[0] - ac2 * ab[0], ab2 * ac[1] - ac2 * ab[1],
                       ab2 * ac[2] - ac2 * ab[2]};
        double w[3] = {u[1] * n[2] - u[2] * n[1], u[2] * n[0] - u[0] * n[2],
                       u[0] * n[1] - u[1] * n[0]};
        for (int i = 0; i < 3; ++i)
            c[i] = p[0][i] + w[i] / (2.0 * n2);
        r2 = std::max(dist2(c, p[0]), std::max(dist2(c, p[1]), dist2(c, p[2])));
    }
    // the stored centre is rounded to float: measure the radius from it
    const float cf[3] = {(float)c[0], (float)c[1], (float)c[2]};
    const double cd[3] = {cf[0], cf[1], cf[2]};
    r2 = std::max(dist2(cd, p[0]), std::max(dist2(cd, p[1]), dist2(cd, p[2])));
    double r = std::sqrt(r2) * 1.001 + 1e-30;
    float r2f = (float)(r * r);
    r2f = std::nextafter(r2f, std::numeric_limits<float>::infinity());
    if (!(r2f == r2f) || !std::isfinite(cf[0]) || !std::isfinite(cf[1]) ||
        !std::isfinite(cf[2]))
        return never_culled_triangle_record();
    return pt::F4{cf[0], cf[1], cf[2], -r2f};
}

} // namespace

// The opaque handle of include/ptgpu.h.
struct ptgpu_scene {
    int device;
    void *blob;          // one allocation: geom | tri | mat | meta | link
    size_t blob_bytes;
    void *bvh_blob = nullptr; // nodes | order | unbounded (large scenes only)
    pt::SceneDesc desc;  // device pointers into blob
    // Which schedule suits this scene, measured on its first large render:
    // -1 not measured yet, 0 pixel-owner kernel, 1 sample-queue kernels.
    // Scheduling only; the image is the same.
    mutable std::atomic<int> queue_mode{-1};
    // Scan form measured best for this scene on the same occasion:
    // -1 not measured, else 0 plain / 1 screened / 2 BVH.
    mutable std::atomic<int> scan_choice{-1};
    mutable std::atomic<int> last_mode{-1}; // schedule the latest render really used
    // Rays per sample measured by the pilots, in thousandths (-1: not
    // measured).  Against the bounce limit it says how much a
    // sample-synchronous wave would waste: nothing when nearly all paths run to
    // the limit (closed scene), and then the pixel-owner kernel may put several
    // lanes on a pixel freely.
    mutable std::atomic<int> path_rays_milli{-1};
    uint64_t content_hash = 0; // of the uploaded records: key of the choice memo
    uint64_t serial = 0;       // unique per upload (never reused): identifies cached tables
};

namespace {
// Choices survive the handle: a caller that uploads the same scene again (the
// one-shot ptgpu_render_frame does, every frame) finds them by content.
struct Choice {
    int queue_mode, scan_choice, path_rays_milli;
};
std::mutex g_choice_mutex;
std::map<uint64_t, Choice> g_choices;

uint64_t fnv1a64(const void *data, size_t n, uint64_t h = 0xCBF29CE484222325ull) {
    const unsigned char *p = static_cast<const unsigned char *>(data);
    for (size_t i = 0; i < n; ++i) {
        h ^= p[i];
        h *= 0x100000001B3ull;
    }
    return h;
}

// Second key: the scene's shape (object / triangle / material counts).  An
// animated scene changes content every frame (the reference's RT loop does,
// src/main.c:260-263) but not shape; it inherits the choice instead of paying
// for pilots on every frame.
uint64_t shape_key(const pt::SceneDesc &d) {
    const uint32_t shape[4] = {d.n_obj, d.n_tri, d.n_mat, d.branching};
    return fnv1a64(shape, sizeof shape, 0x9E3779B97F4A7C15ull);
}

void remember_choice(const ptgpu_scene *s) {
    std::lock_guard<std::mutex> guard(g_choice_mutex);
    if (g_choices.size() > 4096)
        g_choices.clear();
    const Choice c{s->queue_mode.load(), s->scan_choice.load(), s->path_rays_milli.load()};
    g_choices[s->content_hash] = c;
    g_choices[shape_key(s->desc)] = c;
}
} // namespace

extern "C" {

// ---------------------------------------------------------------------------
// parameters
// ---------------------------------------------------------------------------
int ptgpu_params_default(ptgpu_params *p) {
    if (p == nullptr)
        return fail(PTGPU_ERR_INVALID, "params is NULL");
    std::memset(p, 0, sizeof *p);
    p->spp = 16;        // NUM_SAMPLES, src/renderer.c:9
    p->max_bounces = 4; // MAX_BOUNCES, src/renderer.c:8
    // FOV, src/renderer.c:7: M_PI / 180.f * 90.0f (double), used as
    // tanf(FOV / 2.0f); halving commutes with the narrowing to float.
    p->fov_rad = (float)(M_PI / 180.f * 90.0f);
    p->seed = 1;
    p->tile_stride = 1;
    return PTGPU_OK;
}

static bool env_u32(const char *name, uint32_t *out) {
    const char *v = std::getenv(name);
    if (v == nullptr || *v == '\0')
        return false;
    while (*v == ' ' || *v == '\t')
        ++v;
    if (*v == '-') // strtoull would wrap "-1" to 2^64 - 1: not a setting anybody meant
        return false;
    char *end = nullptr;
    unsigned long long x = std::strtoull(v, &end, 0);
    if (end == v || x > 0xFFFFFFFFull)
        return false;
    *out = (uint32_t)x;
    return true;
}

int ptgpu_params_from_env(ptgpu_params *p) {
    int rc = ptgpu_params_default(p);
    if (rc != PTGPU_OK)
        return rc;
    env_u32("PTGPU_SPP", &p->spp);
    env_u32("PTGPU_BOUNCES", &p->max_bounces);
    env_u32("PTGPU_SEED", &p->seed);
    env_u32("PTGPU_BLOCK", &p->block_threads);
    env_u32("PTGPU_FLAGS", &p->flags);
    env_u32("PTGPU_SETTLE", &p->settle_lanes);
    env_u32("PTGPU_LANES_PER_PIXEL", &p->lanes_per_pixel);
    if (const char *deg = std::getenv("PTGPU_FOV_DEG")) {
        char *end = nullptr;
        double d = std::strtod(deg, &end);
        if (end != deg && d > 0.0 && d < 180.0)
            p->fov_rad = (float)(M_PI / 180.0 * d);
    }
    return PTGPU_OK;
}

// ---------------------------------------------------------------------------
// devices
// ---------------------------------------------------------------------------
int ptgpu_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess)
        return fail_hip(e, "hipGetDeviceCount");
    return n;
}

int ptgpu_set_device(int device) {
    PT_TRY_HIP(hipSetDevice(device));
    return PTGPU_OK;
}

// ---------------------------------------------------------------------------
// scene residency
// ---------------------------------------------------------------------------
namespace {
// The spheres that dwarf the rest of a scene: those above the largest gap in the
// sorted radii, if it is a real one (8 x) and at most kGlslBigMax spheres lie
// above it -- configuration 5's ground, the five walls of a Cornell box.  They
// are kept out of the hierarchy and tested for every ray (bvh_build.hpp), and the
// shader-rules mode sizes its box padding by the scene without them (GlslCull).
void spheres_that_dwarf_the_rest(const pt::F4 *geom, uint32_t n_obj, std::vector<char> *out) {
    std::vector<uint32_t> spheres;
    for (uint32_t i = 0; i < n_obj; ++i) {
        const float r2 = geom[i].w;
        if (!std::signbit(r2) && std::isfinite(r2) && std::isfinite(geom[i].x) &&
            std::isfinite(geom[i].y) && std::isfinite(geom[i].z))
            spheres.push_back(i);
    }
    std::sort(spheres.begin(), spheres.end(),
              [&](uint32_t a, uint32_t b) { return geom[a].w > geom[b].w; });
    size_t n_big = 0;
    double widest = 64.0; // (squared radii)
    for (size_t k = 1; k <= pt::kGlslBigMax && k < spheres.size(); ++k) {
        const double above = geom[spheres[k - 1]].w, below = geom[spheres[k]].w;
        const double ratio = below > 0.0 ? above / below : (above > 0.0 ? 1e300 : 0.0);
        if (ratio > widest) {
            widest = ratio;
            n_big = k;
        }
    }
    out->assign(n_obj, 0);
    if (std::getenv("PTGPU_NO_APART") != nullptr) // (experiment knob: everything bounded stays in the tree)
        return;
    for (size_t k = 0; k < n_big; ++k)
        (*out)[spheres[k]] = 1;
}

// pt_layout.h, GlslCull: box and radii of the scene without the objects in `apart`.
void fill_glsl_cull(const pt::F4 *geom, uint32_t n_obj, const std::vector<char> &apart,
                    pt::GlslCull *out) {
    std::memset(out, 0, sizeof *out);
    out->last_scanned = 0xFFFFFFFFu;
    uint32_t last_tri = 0xFFFFFFFFu, last_sph = 0xFFFFFFFFu;
    double lo[3], hi[3];
    for (int a = 0; a < 3; ++a) {
        lo[a] = std::numeric_limits<double>::infinity();
        hi[a] = -lo[a];
    }
    float r2_min = std::numeric_limits<float>::infinity(), r2_max = 0.0f;
    bool any_sphere = false;
    for (uint32_t i = 0; i < n_obj; ++i) {
        const bool tri = std::signbit(geom[i].w);
        (tri ? last_tri : last_sph) = i;
        const double r = std::sqrt((double)std::fabs(geom[i].w));
        const double c[3] = {geom[i].x, geom[i].y, geom[i].z};
        if (apart[i] || !std::isfinite(r) || !std::isfinite(c[0] + c[1] + c[2]))
            continue; // (tested for every ray: no box to pad for them)
        for (int a = 0; a < 3; ++a) {
            lo[a] = std::min(lo[a], c[a] - r);
            hi[a] = std::max(hi[a], c[a] + r);
        }
        if (!tri) {
            any_sphere = true;
            r2_min = std::min(r2_min, geom[i].w);
            r2_max = std::max(r2_max, geom[i].w);
        }
    }
    out->last_scanned = last_tri != 0xFFFFFFFFu ? last_tri : last_sph;
    for (int a = 0; a < 3; ++a) {
        // (outward in binary32; an empty rest leaves an inverted box: no padding needed)
        out->lo[a] = std::nextafter((float)lo[a], -std::numeric_limits<float>::infinity());
        out->hi[a] = std::nextafter((float)hi[a], std::numeric_limits<float>::infinity());
    }
    if (any_sphere) {
        out->r2_max = std::nextafter(r2_max, std::numeric_limits<float>::infinity());
        out->r2_min = r2_min;
    }
}
} // namespace

int ptgpu_scene_upload(const scene *host, ptgpu_scene **out) {
    if (host == nullptr || out == nullptr)
        return fail(PTGPU_ERR_INVALID, "scene or out pointer is NULL");
    *out = nullptr;
    if (host->num_objects > 0x7FFFFFF0u)
        return fail(PTGPU_ERR_LIMIT, "too many objects");
    if (host->num_materials > 65536u)
        return fail(PTGPU_ERR_LIMIT,
                    "more than 65536 materials (16-bit bounce stack)");
    if ((host->num_objects && host->objects == nullptr) ||
        (host->num_materials && host->materials == nullptr))
        return fail(PTGPU_ERR_INVALID, "scene arrays are NULL");

    const uint32_t n_obj = (uint32_t)host->num_objects;
    const uint32_t n_mat = (uint32_t)host->num_materials;
    uint32_t n_tri = 0;
    for (uint32_t i = 0; i < n_obj; ++i) {
        const shape &o = host->objects[i];
        if (o.tag != 0 && o.tag != 1)
            return fail(PTGPU_ERR_INVALID, "object with unknown tag");
        if (o.material >= n_mat)
            return fail(PTGPU_ERR_INVALID, "object refers to a missing material");
        n_tri += (o.tag == 1);
    }

    const size_t bytes = pt::scene_lds_bytes(n_obj, n_tri, n_mat);
    // (+ one record per material behind the part the LDS copy takes: mat_extra)
    std::vector<unsigned char> blob((bytes ? bytes : 16) + (size_t)16 * n_mat, 0);
    pt::F4 *mat_extra = reinterpret_cast<pt::F4 *>(blob.data() + (bytes ? bytes : 16));
    pt::F4 *geom = reinterpret_cast<pt::F4 *>(blob.data());
    pt::F4 *tri = geom + n_obj;
    pt::F4 *mat = tri + 4u * (size_t)n_tri;
    uint32_t *meta = reinterpret_cast<uint32_t *>(mat + 2u * (size_t)n_mat);
    uint32_t *link = meta + n_obj;

    uint32_t k = 0;
    for (uint32_t i = 0; i < n_obj; ++i) {
        const shape &o = host->objects[i];
        if (o.tag == 0) { // SPHERE
            const float r = o.sphere.radius;
            float r2 = r * r;
            if (r2 != r2) // NaN radius: keep it a NaN, but with the sign bit
                r2 = std::numeric_limits<float>::quiet_NaN(); // clear (= sphere)
            geom[i] = pt::F4{o.sphere.center.x, o.sphere.center.y,
                             o.sphere.center.z, r2};
            meta[i] = (uint32_t)o.material;
            link[i] = 0;
        } else { // TRIANGLE
            const pt::V3 v0 = to_v3(o.triangle.v0), v1 = to_v3(o.triangle.v1),
                         v2 = to_v3(o.triangle.v2);
            const pt::V3 e1 = v1 - v0, e2 = v2 - v0;
            // glms_vec3_crossn(v0 - v1, v0 - v2), src/ray.c:50-52
            const pt::V3 n = pt::normalize(pt::cross(v0 - v1, v0 - v2));
            geom[i] = bounding_sphere_record(o.triangle);
            tri[4 * k + 0] = pt::F4{v0.x, v0.y, v0.z, 0.0f};
            tri[4 * k + 1] = pt::F4{e1.x, e1.y, e1.z, 0.0f};
            tri[4 * k + 2] = pt::F4{e2.x, e2.y, e2.z, 0.0f};
            tri[4 * k + 3] = pt::F4{n.x, n.y, n.z, 0.0f};
            meta[i] = (uint32_t)o.material;
            link[i] = k;
            ++k;
        }
    }
    uint32_t branching = 0;
    for (uint32_t m = 0; m < n_mat; ++m) {
        const shape_material &s = host->materials[m];
        // src/renderer.c:37,40: Le = emission_color * emission_strength;
        // deviation scale = roughness * 0.5
        const pt::V3 le = to_v3(s.emission_color) * s.emission_strength;
        mat[2 * m + 0] = pt::F4{s.albedo.x, s.albedo.y, s.albedo.z,
                                (float)((double)s.roughness * 0.5)};
        mat[2 * m + 1] = pt::F4{le.x, le.y, le.z, s.transparency};
        mat_extra[m] = pt::F4{s.roughness, s.refractive_index, 0.0f, 0.0f};
        if (s.transparency > 0.0f && s.transparency < 1.0f)
            branching = 1;
    }

    ptgpu_scene *h = new (std::nothrow) ptgpu_scene();
    if (h == nullptr)
        return fail(PTGPU_ERR_NOMEM, "out of host memory");
    hipError_t e = hipGetDevice(&h->device);
    if (e == hipSuccess)
        e = hipMalloc(&h->blob, blob.size());
    if (e == hipSuccess)
        e = hipMemcpy(h->blob, blob.data(), blob.size(), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        if (h->blob)
            (void)hipFree(h->blob);
        delete h;
        return fail_hip(e, "scene upload");
    }
    h->blob_bytes = blob.size();
    h->content_hash = fnv1a64(blob.data(), blob.size());
    static std::atomic<uint64_t> next_serial{1};
    h->serial = next_serial.fetch_add(1);
    const pt::F4 *dbase = reinterpret_cast<const pt::F4 *>(h->blob);
    h->desc.geom = dbase;
    h->desc.tri = dbase + n_obj;
    h->desc.mat = dbase + n_obj + 4u * (size_t)n_tri;
    h->desc.meta = reinterpret_cast<const uint32_t *>(dbase + n_obj +
                                                      4u * (size_t)n_tri +
                                                      2u * (size_t)n_mat);
    h->desc.link = h->desc.meta + n_obj;
    h->desc.mat_extra = reinterpret_cast<const pt::F4 *>(
        static_cast<const unsigned char *>(h->blob) + (bytes ? bytes : 16));
    h->desc.bvh_nodes = nullptr;
    h->desc.bvh_order = nullptr;
    h->desc.bvh_leaf_geom = nullptr;
    h->desc.bvh_unbounded = nullptr;
    h->desc.n_nodes = 0;
    h->desc.n_unbounded = 0;
    h->desc.bvh_root = pt::kBvhLeafBit;
    h->desc.bvh_depth = 0;
    std::vector<char> apart;
    spheres_that_dwarf_the_rest(geom, n_obj, &apart);
    if (n_obj >= kBvhMinObjects) {
        // triangles: their vertices' extent (much tighter than the box of the
        // bounding sphere the screen uses); spheres: none (lo > hi)
        std::vector<float> corners((size_t)6 * n_obj);
        for (uint32_t i = 0; i < n_obj; ++i) {
            float *ext = &corners[(size_t)6 * i];
            ext[0] = 1.0f;
            ext[3] = 0.0f;
            const shape &o = host->objects[i];
            if (o.tag != 1)
                continue;
            const vec3s v[3] = {o.triangle.v0, o.triangle.v1, o.triangle.v2};
            for (int ax = 0; ax < 3; ++ax) {
                const float x0 = v[0].raw[ax], x1 = v[1].raw[ax], x2 = v[2].raw[ax];
                ext[ax] = std::min(x0, std::min(x1, x2));
                ext[ax + 3] = std::max(x0, std::max(x1, x2));
            }
            if (!(ext[0] <= ext[3])) { // NaN vertex: no extent of its own
                ext[0] = 1.0f;
                ext[3] = 0.0f;
            }
        }
        const pt::BvhHost bvh = pt::build_bvh(geom, n_obj, corners.data(), apart.data());
        // A leaf visit fetches kLeafMax slots of `order` and of the geometry
        // copied into leaf order, whatever the leaf's count (one memory round
        // trip instead of two per object): both arrays are padded accordingly.
        const size_t slots = bvh.order.size() + pt::kLeafMax;
        std::vector<pt::F4> leaf_geom(slots, pt::F4{0.0f, 0.0f, 0.0f, 0.0f});
        for (size_t j = 0; j < bvh.order.size(); ++j)
            leaf_geom[j] = geom[bvh.order[j]];
        const std::vector<pt::F4> packed_nodes = pt::pack_nodes_half(bvh.nodes);
        const size_t node_bytes = packed_nodes.size() * sizeof(pt::F4);
        const size_t leaf_bytes = slots * sizeof(pt::F4);
        const size_t order_bytes = slots * sizeof(uint32_t);
        const size_t unb_bytes = bvh.unbounded.size() * sizeof(uint32_t);
        const size_t total = node_bytes + leaf_bytes + order_bytes + unb_bytes + 16;
        std::vector<unsigned char> packed(total, 0);
        if (node_bytes) // (memcpy's arguments may not be null, even for 0 bytes)
            std::memcpy(packed.data(), packed_nodes.data(), node_bytes);
        std::memcpy(packed.data() + node_bytes, leaf_geom.data(), leaf_bytes);
        if (!bvh.order.empty())
            std::memcpy(packed.data() + node_bytes + leaf_bytes, bvh.order.data(),
                        bvh.order.size() * sizeof(uint32_t));
        if (unb_bytes)
            std::memcpy(packed.data() + node_bytes + leaf_bytes + order_bytes, bvh.unbounded.data(), unb_bytes);
        hipError_t be = hipMalloc(&h->bvh_blob, total);
        if (be == hipSuccess)
            be = hipMemcpy(h->bvh_blob, packed.data(), total, hipMemcpyHostToDevice);
        if (be != hipSuccess) {
            if (h->bvh_blob)
                (void)hipFree(h->bvh_blob);
            (void)hipFree(h->blob);
            delete h;
            return fail_hip(be, "hierarchy upload");
        }
        const unsigned char *b0 = static_cast<const unsigned char *>(h->bvh_blob);
        h->desc.bvh_nodes = reinterpret_cast<const pt::F4 *>(b0);
        h->desc.bvh_leaf_geom = reinterpret_cast<const pt::F4 *>(b0 + node_bytes);
        h->desc.bvh_order = reinterpret_cast<const uint32_t *>(b0 + node_bytes + leaf_bytes);
        h->desc.bvh_unbounded = reinterpret_cast<const uint32_t *>(b0 + node_bytes + leaf_bytes + order_bytes);
        h->desc.n_nodes = (uint32_t)(bvh.nodes.size() / 4);
        h->desc.n_unbounded = (uint32_t)bvh.unbounded.size();
        h->desc.bvh_root = bvh.root;
        h->desc.bvh_depth = bvh.depth;
    }
    h->desc.n_obj = n_obj;
    h->desc.n_tri = n_tri;
    h->desc.n_mat = n_mat;
    h->desc.branching = branching;
    h->desc.sky[0] = host->sky_color.x;
    h->desc.sky[1] = host->sky_color.y;
    h->desc.sky[2] = host->sky_color.z;
    h->desc.last_sphere = 0xFFFFFFFFu;
    for (uint32_t i = 0; i < n_obj; ++i)
        if (host->objects[i].tag == 0)
            h->desc.last_sphere = i;
    fill_glsl_cull(geom, n_obj, apart, &h->desc.glsl);
    {
        std::lock_guard<std::mutex> guard(g_choice_mutex);
        auto known = g_choices.find(h->content_hash);
        if (known == g_choices.end())
            known = g_choices.find(shape_key(h->desc));
        if (known != g_choices.end()) {
            h->queue_mode.store(known->second.queue_mode);
            h->scan_choice.store(known->second.scan_choice);
            h->path_rays_milli.store(known->second.path_rays_milli);
            if (h->scan_choice.load() == 2 && h->desc.bvh_nodes == nullptr)
                h->scan_choice.store(-1); // inherited by shape, but no hierarchy here
        }
    }
    *out = h;
    return PTGPU_OK;
}

void ptgpu_scene_free(ptgpu_scene *s) {
    if (s == nullptr)
        return;
    if (s->blob)
        (void)hipFree(s->blob);
    if (s->bvh_blob)
        (void)hipFree(s->bvh_blob);
    delete s;
}

int ptgpu_scene_schedule(const ptgpu_scene *s) {
    if (s == nullptr)
        return fail(PTGPU_ERR_INVALID, "scene is NULL");
    // what the latest render of this scene used (it can differ from the
    // remembered per-scene choice: small row sets go to the queue regardless)
    const int used = s->last_mode.load();
    return (used >= 0 ? used : s->queue_mode.load()) + 1;
}

int ptgpu_scene_scan(const ptgpu_scene *s) {
    if (s == nullptr)
        return fail(PTGPU_ERR_INVALID, "scene is NULL");
    return s->scan_choice.load() + 1; // 0 not measured, 1 plain, 2 screened, 3 BVH
}

int ptgpu_scene_hierarchy(const ptgpu_scene *s, uint32_t *nodes, uint32_t *depth) {
    if (s == nullptr || nodes == nullptr || depth == nullptr)
        return fail(PTGPU_ERR_INVALID, "NULL argument");
    *nodes = s->desc.bvh_nodes != nullptr ? s->desc.n_nodes : 0u;
    *depth = s->desc.bvh_nodes != nullptr ? s->desc.bvh_depth : 0u;
    return PTGPU_OK;
}

int ptgpu_scene_is_branching(const ptgpu_scene *s) {
    if (s == nullptr)
        return fail(PTGPU_ERR_INVALID, "scene is NULL");
    return (int)s->desc.branching;
}

// ---------------------------------------------------------------------------
// rendering
// ---------------------------------------------------------------------------
namespace {

// Settle threshold from the scene size.  One pass of the bounce loop costs
// about 80 VALU instructions per object scanned plus ~400 for shading; one
// settle stage about 265 (at 8 bounces) whatever the lane count.  Parking f
// lanes for a pass wastes f/64 of a pass; settling costs one settle stage:
// settle as soon as f * pass >= 64 * settle.
// Stream kernel (sample queue + BVH scan): a service stage runs when this many
// lanes of a wave wait for one, and a walk quantum takes this many inner-node
// steps at most (PTGPU_WALK_BUDGET overrides, for tuning).
constexpr uint32_t kStreamServiceLanes = 32;
// ... and shades a second time within the stage when the refill has handed out
// this many tasks that start at a known primary hit (C5, 32 spp: 20.4 -> 19.4 ms)
constexpr uint32_t kStreamReshadeLanes = 4;
// Between full services, 16 lanes with a finished walk get a shading-only stage
// (finished walks are most of what waits; C5, 32 spp: 17.6 -> 16.7 ms)
constexpr uint32_t kStreamShadeLanes = 16;
// (PTGPU_LEAF_LANES / PTGPU_RESOLVE_LANES: tuning knobs of the same kind)
uint32_t stream_knob(const char *name, uint32_t fallback) {
    uint32_t v = fallback;
    if (env_u32(name, &v) && v >= 1 && v <= 64)
        return v;
    return fallback;
}
uint32_t stream_walk_budget() {
    static const uint32_t budget = [] {
        uint32_t v = 5u; // (round 3, after the inner step got cheaper: 3..12 measured on C5 --
                         // 37.6 / 36.9 / 35.4 / 36.4 / 37.0 / 38.0 / 37.2 / 38.2 ms per 64 spp)
        if (const char *e = std::getenv("PTGPU_WALK_BUDGET")) {
            const long x = std::strtol(e, nullptr, 10);
            if (x >= 1 && x <= 1024)
                v = (uint32_t)x;
        }
        return v;
    }();
    return budget;
}

uint32_t auto_settle_lanes(const pt::SceneDesc &d) {
    const double pass = 80.0 * (double)d.n_obj + 400.0;
    double f = 64.0 * 265.0 / pass;
    if (f < 1.0)
        f = 1.0;
    if (f > 64.0)
        f = 64.0;
    return (uint32_t)(f + 0.5);
}

// Validates a call and fills the launch block (everything but the
// destination pointers).
int prepare_launch(const ptgpu_scene *s, size_t framew, size_t frameh,
                   const ptgpu_params *p, pt::LaunchArgs *a, uint32_t *block,
                   bool *tree) {
    if (s == nullptr || p == nullptr)
        return fail(PTGPU_ERR_INVALID, "scene or params is NULL");
    if (framew == 0 || frameh == 0 || framew > 0xFFFFu * 16u ||
        frameh > 0xFFFFu * 16u || framew * frameh > 0xFFFFFFFFull)
        return fail(PTGPU_ERR_INVALID, "frame size out of range");
    if (p->spp == 0)
        return fail(PTGPU_ERR_INVALID, "spp must be >= 1");
    if (p->row_begin >= frameh)
        return fail(PTGPU_ERR_INVALID, "row_begin is below the frame");

    uint32_t num_rows = p->num_rows;
    const uint32_t stride = p->tile_stride ? p->tile_stride : 1u;
    if (num_rows == 0) {
        if (p->tile_rows != 0 && stride != 1)
            return fail(PTGPU_ERR_INVALID,
                        "num_rows must be given for a strided tile set");
        num_rows = (uint32_t)frameh - p->row_begin;
    }
    // last global row of the set must exist
    uint64_t last = (uint64_t)p->row_begin + (num_rows - 1u);
    if (p->tile_rows != 0) {
        const uint64_t tile = (num_rows - 1u) / p->tile_rows;
        last = (uint64_t)p->row_begin + tile * p->tile_rows * stride +
               ((num_rows - 1u) % p->tile_rows);
    }
    if (last >= frameh)
        return fail(PTGPU_ERR_INVALID, "row set reaches below the frame");

    *tree = s->desc.branching || (p->flags & PTGPU_FLAG_FORCE_TREE);
    if (*tree && p->max_bounces > PTGPU_MAX_BOUNCES_TREE)
        return fail(PTGPU_ERR_LIMIT,
                    "max_bounces exceeds the branching kernel's frame stack");
    if (!*tree && p->max_bounces > PTGPU_MAX_BOUNCES_PATH)
        return fail(PTGPU_ERR_LIMIT, "max_bounces exceeds the kernel's limit");

    const bool want_lds = (p->flags & PTGPU_FLAG_SCENE_LDS) &&
                          !(p->flags & PTGPU_FLAG_SCENE_GLOBAL);
    uint32_t b = p->block_threads ? p->block_threads : (want_lds ? 256u : 64u);
    if (b % 64u != 0 || b > 1024u)
        return fail(PTGPU_ERR_INVALID,
                    "block_threads must be a multiple of 64, at most 1024");
    *block = b;

    std::memset(a, 0, sizeof *a);
    a->scene = s->desc;
    a->framew = (uint32_t)framew;
    a->frameh = (uint32_t)frameh;
    a->num_rows = num_rows;
    a->row_begin = p->row_begin;
    a->tile_rows = p->tile_rows;
    a->tile_stride = stride;
    a->compact = (p->flags & PTGPU_FLAG_COMPACT_ROWS) ? 1u : 0u;
    a->spp = p->spp;
    a->sample_offset = p->sample_offset;
    a->max_bounces = p->max_bounces;
    a->seed = p->seed;
    a->tan_half_fov = tanf(p->fov_rad / 2.0f); // src/renderer.c:77
    a->scene_in_lds = want_lds ? 1u : 0u;
    // scan form: what was measured for this scene, else by object count
    // (hit density matters more than count -- 8 wall-sized spheres want the
    // plain scan, 8 small ones the screened scan -- which is why it is
    // measured when a render is large enough, see render_common)
    a->screened = (s->desc.n_obj >= 32u) ? 1u : 0u;
    if (s->desc.n_obj >= kBvhAutoObjects && s->desc.bvh_nodes != nullptr)
        a->screened = 2u;
    if (s->scan_choice.load() >= 0)
        a->screened = (uint32_t)s->scan_choice.load();
    if (p->flags & PTGPU_FLAG_SCAN_PLAIN)
        a->screened = 0u;
    if (p->flags & PTGPU_FLAG_SCAN_SCREENED)
        a->screened = 1u;
    if (p->flags & PTGPU_FLAG_SCAN_BVH) {
        if (s->desc.bvh_nodes == nullptr && s->desc.n_unbounded == 0)
            return fail(PTGPU_ERR_INVALID, "scene has no hierarchy (fewer than 64 objects)");
        a->screened = 2u;
    }
    a->share_primary = (p->flags & PTGPU_FLAG_NO_PRIMARY_SHARE) ? 0u : 1u;
    if (p->lanes_per_pixel > 64u || (p->lanes_per_pixel & (p->lanes_per_pixel - 1u)) != 0u)
        return fail(PTGPU_ERR_INVALID, "lanes_per_pixel must be 0 or a power of two up to 64");
    a->lanes_per_pixel = 1u; // render_common decides
    a->stream_walks = 0u;
    a->tree = *tree ? 1u : 0u;
    a->chunk_batch = 1u; // the launcher decides
    a->primary_hits = nullptr;
    a->walk_budget = 0u;
    if (p->settle_lanes > 64u)
        return fail(PTGPU_ERR_INVALID, "settle_lanes must be 0..64");
    a->settle_lanes = p->settle_lanes ? p->settle_lanes : auto_settle_lanes(s->desc);
    if (a->scene_in_lds &&
        pt::megakernel_lds_bytes(*a, b, *tree) > kLdsBudget)
        a->scene_in_lds = 0; // too large for one CU's LDS: scan from L2/HBM
    if (pt::megakernel_lds_bytes(*a, b, *tree) > kLdsBudget)
        return fail(PTGPU_ERR_LIMIT, "bounce stack does not fit LDS; "
                                     "lower max_bounces or block_threads");

    a->counters = nullptr;
    if (!(p->flags & PTGPU_FLAG_NO_RAY_COUNT)) {
        int rc = counters_for_current_device(&a->counters);
        if (rc != PTGPU_OK)
            return rc;
    }
    return PTGPU_OK;
}

// The two builds of the device code (csrc/device/pt_vec.h): bit-exact, and the
// opt-in tolerance mode (PTGPU_FLAG_FAST_MATH).
struct Launchers {
    decltype(&pt::launch_megakernel) megakernel;
    decltype(&pt::launch_samplekernel) samplekernel;
    decltype(&pt::launch_sum) sum;
    decltype(&pt::launch_primary) primary;
    decltype(&pt::launch_trace) trace;
};
const Launchers kExactBuild{&pt::launch_megakernel, &pt::launch_samplekernel, &pt::launch_sum,
                            &pt::launch_primary, &pt::launch_trace};
const Launchers kFastBuild{&pt_fast::launch_megakernel, &pt_fast::launch_samplekernel,
                           &pt_fast::launch_sum, &pt_fast::launch_primary,
                           &pt_fast::launch_trace};
const Launchers *launchers_for(uint32_t flags) {
    return (flags & PTGPU_FLAG_FAST_MATH) ? &kFastBuild : &kExactBuild;
}

// launches are counted on the host (per device)
std::atomic<unsigned long long> g_launches[kMaxDevices];

void count_launch() {
    int dev = 0;
    if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < kMaxDevices)
        g_launches[dev].fetch_add(1, std::memory_order_relaxed);
}

} // namespace

namespace {

// Per-device scratch of the sample-queue mode, grown on demand.
// What a primary-hit table was computed for: the upload (by serial, exact),
// the camera and the row set.
struct PrimaryView {
    uint64_t scene_serial = 0;
    uint32_t framew = 0, frameh = 0, row_begin = 0, num_rows = 0, tile_rows = 0, tile_stride = 0;
    uint32_t tan_half_fov_bits = 0;
    uint32_t fast_math = 0; // which build of the device code computed the table
    bool operator==(const PrimaryView &o) const {
        return fast_math == o.fast_math && scene_serial == o.scene_serial && framew == o.framew && frameh == o.frameh &&
               row_begin == o.row_begin && num_rows == o.num_rows && tile_rows == o.tile_rows &&
               tile_stride == o.tile_stride && tan_half_fov_bits == o.tan_half_fov_bits;
    }
};

struct DeviceScratch {
    float *samples = nullptr;   // float4 per task of one pass
    size_t sample_bytes = 0;
    float *accum = nullptr;     // float4 per pixel (fused renders)
    size_t accum_bytes = 0;
    float *primary = nullptr;   // (distance, object) per pixel: primary-ray hits
    size_t primary_bytes = 0;
    PrimaryView primary_view;   // what `primary` holds (serial 0: nothing)
    uint32_t *chunk_counters = nullptr; // ring of kCounterRing counters
    uint32_t next_counter = 0;
    uint32_t max_waves = 0;     // waves the device keeps resident
    hipEvent_t last_use = nullptr; // end of the last frame that used the scratch
    std::mutex lock;            // held while a frame's launches are enqueued
};
constexpr uint32_t kCounterRing = 256;
DeviceScratch g_scratch[kMaxDevices];

size_t scratch_budget_bytes() {
    uint32_t mb = 2048; // sample buffer budget per device
    env_u32("PTGPU_SCRATCH_MB", &mb);
    if (mb < 64)
        mb = 64;
    return (size_t)mb << 20;
}

int grow(float **ptr, size_t *have, size_t need) {
    if (*have >= need)
        return PTGPU_OK;
    PT_TRY_HIP(hipDeviceSynchronize()); // nothing may still be using the old one
    if (*ptr)
        (void)hipFree(*ptr);
    *ptr = nullptr;
    *have = 0;
    PT_TRY_HIP(hipMalloc(ptr, need));
    *have = need;
    return PTGPU_OK;
}

int scratch_for_current_device(DeviceScratch **out) {
    int dev = 0;
    int rc = current_device(&dev);
    if (rc != PTGPU_OK)
        return rc;
    DeviceScratch &sc = g_scratch[dev];
    std::lock_guard<std::mutex> guard(g_ctx_mutex);
    if (sc.chunk_counters == nullptr) {
        PT_TRY_HIP(hipMalloc(&sc.chunk_counters, kCounterRing * sizeof(uint32_t)));
        PT_TRY_HIP(hipEventCreateWithFlags(&sc.last_use, hipEventDisableTiming));
        hipDeviceProp_t prop;
        PT_TRY_HIP(hipGetDeviceProperties(&prop, dev));
        const int per_cu = prop.maxThreadsPerMultiProcessor / 64;
        sc.max_waves = (uint32_t)(prop.multiProcessorCount * (per_cu > 0 ? per_cu : 32));
    }
    *out = &sc;
    return PTGPU_OK;
}

// Device time and algorithmic bytes of the render calls, for ptgpu_stats.  A
// call brackets its kernels with two events on its stream; pairs are turned
// into milliseconds once they have completed (never waited for).
struct DeviceTiming {
    std::mutex lock;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
    std::vector<hipEvent_t> spare;
    double ms = 0.0;
    unsigned long long bytes = 0;

    hipEvent_t take() { // lock held
        if (!spare.empty()) {
            hipEvent_t e = spare.back();
            spare.pop_back();
            return e;
        }
        hipEvent_t e = nullptr;
        return hipEventCreate(&e) == hipSuccess ? e : nullptr;
    }
    void harvest() {
        std::lock_guard<std::mutex> guard(lock);
        size_t kept = 0;
        for (auto &pr : pending) {
            float t = 0.0f;
            if (hipEventQuery(pr.second) == hipSuccess &&
                hipEventElapsedTime(&t, pr.first, pr.second) == hipSuccess) {
                ms += t;
                spare.push_back(pr.first);
                spare.push_back(pr.second);
            } else {
                pending[kept++] = pr;
            }
        }
        pending.resize(kept);
        (void)hipGetLastError(); // hipEventQuery reports "not ready" as an error
    }
};
DeviceTiming g_timing[kMaxDevices];

// Brackets one call's work on `stream`.
struct TimedCall {
    DeviceTiming *t = nullptr;
    hipStream_t stream;
    hipEvent_t begin = nullptr, end = nullptr;
    TimedCall(bool enabled, hipStream_t st) : stream(st) {
        int dev = 0;
        if (!enabled || current_device(&dev) != PTGPU_OK)
            return;
        t = &g_timing[dev];
        std::lock_guard<std::mutex> guard(t->lock);
        begin = t->take();
        end = t->take();
        if (begin == nullptr || end == nullptr || hipEventRecord(begin, stream) != hipSuccess)
            t = nullptr;
    }
    void add_bytes(unsigned long long n) {
        if (t != nullptr) {
            std::lock_guard<std::mutex> guard(t->lock);
            t->bytes += n;
        }
    }
    ~TimedCall() {
        if (t == nullptr)
            return;
        bool many = false;
        {
            std::lock_guard<std::mutex> guard(t->lock);
            if (hipEventRecord(end, stream) == hipSuccess)
                t->pending.emplace_back(begin, end);
            many = t->pending.size() > 64;
        }
        if (many)
            t->harvest();
    }
};

enum Mode { kPixelOwner = 0, kSampleQueue = 1 };

// One frame's worth of work for both public entry points.  Exactly one of
// `d_frame` (fused render: RGB8 out) and `d_accum` (caller's accumulator) is
// non-null.  Every path below adds the samples of a pixel in sample order, so
// they all produce the same bits; they differ in schedule only.
// `may_pilot`: an unmeasured scene may be measured inside this call (timed pilot
// passes that synchronise the stream a few times).  The synchronous entry
// points (host frames, the legacy shims, ptgpu_scene_tune) allow it; the
// asynchronous device-frame entry points do not, and go by object count until
// the scene has been tuned.
int render_common(const ptgpu_scene *s, uint8_t *d_frame, float *d_accum,
                  size_t framew, size_t frameh, const ptgpu_params *p,
                  hipStream_t stream, bool may_pilot) {
    pt::LaunchArgs a;
    uint32_t block = 0;
    bool tree = false;
    int rc = prepare_launch(s, framew, frameh, p, &a, &block, &tree);
    if (rc != PTGPU_OK)
        return rc;
    const uint64_t npix = (uint64_t)a.num_rows * a.framew;
    const uint32_t total_spp = a.spp;
    if (p->flags & PTGPU_FLAG_GLSL_SEMANTICS) {
        // the shader's rules (pt_glsl.hip): its random stream is one state per
        // pixel carried from sample to sample, so a frame is one launch of all
        // its samples, from sample 0, straight to RGB8
        if (d_frame == nullptr)
            return fail(PTGPU_ERR_INVALID, "GLSL semantics: no accumulator form (the shader's "
                                           "stream is carried across a pixel's samples)");
        if (p->sample_offset != 0)
            return fail(PTGPU_ERR_INVALID, "GLSL semantics: sample_offset must be 0");
        if (p->max_bounces > pt::kGlslMaxBounces)
            return fail(PTGPU_ERR_LIMIT, "GLSL semantics: max_bounces exceeds the ray stack");
        if (p->flags & PTGPU_FLAG_FAST_MATH)
            return fail(PTGPU_ERR_INVALID, "GLSL semantics has no tolerance build");
        TimedCall timed_glsl(a.counters != nullptr, stream);
        a.frame = d_frame;
        const hipError_t e = pt::launch_glsl(a, stream);
        if (e != hipSuccess)
            return fail_hip(e, "GLSL-semantics launch");
        count_launch();
        timed_glsl.add_bytes(3ull * npix);
        return PTGPU_OK;
    }
    const Launchers &L = *launchers_for(p->flags);
    TimedCall timed(a.counters != nullptr, stream); // part of the statistics

    // ---- which schedule ------------------------------------------------------
    int mode = s->queue_mode.load();
    if (p->flags & PTGPU_FLAG_PIXEL_OWNER)
        mode = kPixelOwner;
    if (p->flags & PTGPU_FLAG_SAMPLE_QUEUE)
        mode = kSampleQueue;
    const bool can_measure = may_pilot && a.counters != nullptr && total_spp >= 16 &&
                             npix * total_spp >= (1ull << 22);
    if (mode < 0 && !can_measure) // (the BVH scan is at its best in queue form)
        mode = a.screened == 2u ? kSampleQueue : kPixelOwner;

    // How many lanes share a pixel in the pixel-owner kernel (pt_kernels.hip,
    // run_paths_split).  More lanes = more, shorter waves: a small row set (a
    // shard of a frame split over many GPUs, a small image) fills the device,
    // and a large one ends with a shorter tail of half-empty wave rounds
    // (measured, configuration 2: 63.3 ms with 1 lane, 59.0 with 8; a 135-row
    // shard 11.9 -> 8.9).  The price: such waves are sample-synchronous, which
    // wastes lane-passes when path lengths differ.  So: as many lanes as it
    // takes to reach ~24 rounds of waves when the scene is known to run its
    // paths to the bounce limit, and otherwise only as many as fit into
    // wave slots that would stay empty.
    uint32_t owner_block = block; // workgroup size of pixel-owner launches
    auto choose_lanes = [&](bool may_fall_back_to_queue) {
        owner_block = block;
        if (mode != kPixelOwner)
            return;
        DeviceScratch *probe = nullptr;
        if (p->lanes_per_pixel != 0u) {
            a.lanes_per_pixel = p->lanes_per_pixel;
        } else if (scratch_for_current_device(&probe) == PTGPU_OK && probe->max_waves > 0) {
            const double fill = (double)((npix + 63u) / 64u) / (double)probe->max_waves;
            // average path length over the longest possible one = the lane use
            // of a sample-synchronous wave, roughly.  Near 1 (closed scene):
            // ~24 rounds of waves.  From 0.4 (open scene, short bounce limit:
            // the reference's own scene at its 4 bounces is 0.44): ~16 rounds,
            // where the finer granularity still beats the idle lane-passes
            // (1280x800: 0.65 -> 0.56 ms with 4 lanes; 1080p 1.04 -> 0.99 with
            // 2).  Below that (the same scene at 8 bounces: 0.25) or unknown:
            // only wave slots that would stay empty.
            const int milli = s->path_rays_milli.load();
            const double length_ratio =
                milli < 0 ? 0.0 : std::min(1.0, 1e-3 * milli / ((double)a.max_bounces + 1.0));
            const double want = length_ratio >= 0.93 ? 24.0 : length_ratio >= 0.4 ? 16.0 : 1.0;
            uint32_t k = 1u;
            while (fill * (2u * k) <= want * 1.34 && k < 16u && 2u * k <= total_spp)
                k *= 2u;
            a.lanes_per_pixel = k;
            // Still under-filled with one lane per pixel (open scene, few
            // samples): the sample queue fills the device whatever the row
            // set, at the price of 32 bytes per sample.  Measured on C2
            // shards: one round of waves (fill ~1) is 11 % slower than the
            // queue, half a round 27 %, two rounds equal.
            const double owner_efficiency = fill / std::ceil(fill);
            if (may_fall_back_to_queue && k == 1u && a.counters != nullptr && total_spp >= 16 &&
                !(p->flags & PTGPU_FLAG_PIXEL_OWNER) &&
                (fill < 1.5 || (fill < 4.0 && owner_efficiency < 0.85))) {
                mode = kSampleQueue;
                return;
            }
        }
        // With k lanes per pixel: workgroups of up to 256 work-items, so that a
        // workgroup's pixels leave as one run of whole dwords
        // (store_rgb8_block; 192 bytes for k <= 4) and its statistics as one
        // atomic.  Larger workgroups cost time (C2, k = 8: 62.1 ms at 64..256
        // work-items, 66.4 at 512).
        // (a launch of less than two rounds of waves keeps single-wave
        // workgroups, which spread better: the 256x256 C1 frame 0.054 vs 0.063 ms)
        const uint32_t wanted_block = std::min(256u, 64u * a.lanes_per_pixel);
        const uint64_t launch_waves = (npix * a.lanes_per_pixel + 63u) / 64u;
        DeviceScratch *dev = nullptr;
        const bool large = scratch_for_current_device(&dev) == PTGPU_OK && dev->max_waves > 0 &&
                           launch_waves >= 2ull * dev->max_waves;
        if (a.lanes_per_pixel > 1u && p->block_threads == 0u && wanted_block > block && large &&
            pt::megakernel_lds_bytes(a, wanted_block, false) <= 64u * 1024u)
            owner_block = wanted_block;
    };
    choose_lanes(true);

    if (mode >= 0)
        s->last_mode.store(mode);
    if (mode == kPixelOwner) { // the one-launch fast path, no scratch involved
        a.frame = d_frame;  // exactly one of the two is non-null
        a.accum = d_accum;
        a.accum_local = 0;
        hipError_t e = L.megakernel(a, owner_block, tree, stream);
        if (e != hipSuccess)
            return fail_hip(e, "megakernel launch");
        count_launch();
        timed.add_bytes((d_frame != nullptr ? 3ull : 32ull) * npix);
        return PTGPU_OK;
    }

    // ---- everything else runs through an accumulator ----------------------------
    DeviceScratch *sc = nullptr;
    rc = scratch_for_current_device(&sc);
    if (rc != PTGPU_OK)
        return rc;
    // The scratch buffers are shared by every frame on this device: enqueue
    // one frame at a time, and make this stream wait for the previous frame
    // that used them (which may have run on another stream).
    std::lock_guard<std::mutex> scratch_guard(sc->lock);
    PT_TRY_HIP(hipStreamWaitEvent(stream, sc->last_use, 0));
    struct MarkUse {
        DeviceScratch *sc;
        hipStream_t stream;
        ~MarkUse() { (void)hipEventRecord(sc->last_use, stream); }
    } mark_use{sc, stream};
    float *accum = d_accum;
    a.accum_local = 0;
    bool accum_is_zero = false; // scratch accumulator not written yet
    if (d_frame != nullptr) {
        rc = grow(&sc->accum, &sc->accum_bytes, (size_t)npix * 16u);
        if (rc != PTGPU_OK)
            return rc;
        accum = sc->accum;
        a.accum_local = 1;
        accum_is_zero = true;
    }
    a.frame = nullptr;
    a.accum = accum;
    uint32_t done_spp = 0;

    // One pass of the sample-queue schedule: the next `want` samples (or as
    // many as the sample buffer holds) of every pixel traced by persistent
    // waves into the buffer, then added to the accumulator in sample order;
    // the pass that completes the frame also writes RGB8.
    const uint32_t npix_pad = (uint32_t)((npix + 63u) & ~(uint64_t)63u);
    bool primary_ready = false; // sc->primary holds this call's primary-ray hits
    auto queue_pass = [&](uint32_t scan, uint32_t want) -> int {
        uint64_t per_pass = scratch_budget_bytes() / ((uint64_t)npix_pad * 16u);
        if (per_pass < 1)
            per_pass = 1;
        if ((uint64_t)npix_pad * per_pass > 0xFFFFFFFFull) // task ids are 32-bit
            per_pass = 0xFFFFFFFFull / npix_pad;
        const uint32_t k = (uint32_t)(want < per_pass ? want : per_pass);
        // (sized for the frame's largest pass at once, not pass by pass)
        const uint64_t k_frame = total_spp < per_pass ? total_spp : per_pass;
        int grc = grow(&sc->samples, &sc->sample_bytes, (size_t)npix_pad * k_frame * 16u);
        if (grc != PTGPU_OK)
            return grc;
        pt::LaunchArgs q = a;
        q.screened = scan;
        q.spp = k;
        q.sample_offset = a.sample_offset + done_spp;
        q.sample_buf = sc->samples;
        q.npix_pad = npix_pad;
        q.chunks_per_sample = npix_pad / 64u;
        q.num_chunks = k * q.chunks_per_sample;
        q.chunk_counter = sc->chunk_counters + (sc->next_counter++ % kCounterRing);
        PT_TRY_HIP(hipMemsetAsync(q.chunk_counter, 0, sizeof(uint32_t), stream));
        if (scan == 2u && !(p->flags & PTGPU_FLAG_WHOLE_WALKS)) {
            // every sample of a pixel starts with the same primary ray: its
            // hit is found once per call and handed to the tasks
            if (a.share_primary && !primary_ready) {
                // (kept from call to call: progressive passes of one view find
                // the table of the previous pass)
                const PrimaryView view{s->serial, a.framew, a.frameh, a.row_begin, a.num_rows,
                                       a.tile_rows, a.tile_stride,
                                       __builtin_bit_cast(uint32_t, a.tan_half_fov),
                                       (p->flags & PTGPU_FLAG_FAST_MATH) ? 1u : 0u};
                if (!(sc->primary_view == view) || sc->primary_bytes < (size_t)npix * 8u) {
                    sc->primary_view = PrimaryView{};
                    grc = grow(&sc->primary, &sc->primary_bytes, (size_t)npix * 8u);
                    if (grc != PTGPU_OK)
                        return grc;
                    hipError_t pe = L.primary(a, sc->primary, stream);
                    if (pe != hipSuccess)
                        return fail_hip(pe, "primary-hit launch");
                    sc->primary_view = view;
                }
                primary_ready = true;
            }
            if (a.share_primary)
                q.primary_hits = sc->primary;
            // resumable walks: settle_lanes = lanes waiting before a service stage
            q.stream_walks = 1u;
            if (!*&a.tree && !a.scene_in_lds && block == 64u) { // two paths per lane (pt_duo.h)
                const char *duo = std::getenv("PTGPU_DUO");
                if (duo != nullptr && duo[0] == '1')
                    q.stream_walks = 2u;
            }
            q.settle_lanes = p->settle_lanes ? p->settle_lanes : kStreamServiceLanes;
            q.walk_budget = stream_walk_budget();
            static const uint32_t leaf_lanes = stream_knob("PTGPU_LEAF_LANES", 16u);
            static const uint32_t resolve_lanes = stream_knob("PTGPU_RESOLVE_LANES", 4u);
            q.leaf_lanes = leaf_lanes;
            q.resolve_lanes = resolve_lanes;
            static const uint32_t shade_lanes = stream_knob("PTGPU_SHADE_LANES", kStreamShadeLanes);
            static const uint32_t reshade_lanes = stream_knob("PTGPU_RESHADE_LANES", kStreamReshadeLanes);
            static const uint32_t service_lanes = stream_knob("PTGPU_SERVICE_LANES", kStreamServiceLanes);
            q.reshade_lanes = reshade_lanes;
            q.shade_lanes = shade_lanes;
            if (!p->settle_lanes)
                q.settle_lanes = service_lanes;
            if (q.stream_walks == 2u) { // (tuning knobs, read per call)
                auto knob = [](const char *name, uint32_t fallback) {
                    const char *e = std::getenv(name);
                    if (e == nullptr)
                        return fallback;
                    const long x = std::strtol(e, nullptr, 10);
                    return (x >= 1 && x <= 1024) ? (uint32_t)x : fallback;
                };
                q.settle_lanes = p->settle_lanes ? p->settle_lanes : knob("PTGPU_DUO_SERVICE", kStreamServiceLanes);
                q.shade_lanes = knob("PTGPU_DUO_SHADE", q.shade_lanes);
                q.reshade_lanes = knob("PTGPU_DUO_RESHADE", q.reshade_lanes);
                q.leaf_lanes = knob("PTGPU_DUO_LEAF", q.leaf_lanes);
                q.resolve_lanes = knob("PTGPU_DUO_RESOLVE", q.resolve_lanes);
                q.walk_budget = knob("PTGPU_DUO_BUDGET", q.walk_budget);
            }
        }
        hipError_t e = L.samplekernel(q, block, sc->max_waves, stream);
        if (e != hipSuccess)
            return fail_hip(e, "sample kernel launch");
        count_launch();
        // samples stored and read back in order; accumulator read (unless this
        // is its first pass) and written, or the frame written
        timed.add_bytes(32ull * npix * k + (accum_is_zero ? 0ull : 16ull * npix) + 16ull * npix);
        done_spp += k;
        q.first_pass = accum_is_zero ? 1u : 0u;
        accum_is_zero = false;
        q.frame = (d_frame != nullptr && done_spp == total_spp) ? d_frame : nullptr;
        e = L.sum(q, stream);
        if (e != hipSuccess)
            return fail_hip(e, "sum kernel launch");
        return PTGPU_OK;
    };

    // ---- first large render of this scene: measure, then choose ----------------
    // Pilots = the first samples of the frame (they count: every scan form and
    // schedule gives the same samples), rendered once per candidate scan form
    // untimed (code-object load, clocks) and then in two interleaved timed
    // rounds (A B A B), so that drift does not favour whoever runs second.  A
    // linear scan is piloted in pixel-owner form; the BVH scan in the form it
    // would really run in, sample-queue passes with resumable walks.  The
    // fastest form is kept for this scene.  For a linear scan the pilots also
    // yield rays per sample and the lane use of pixel-owner waves: the queue
    // schedule recovers the idle lane-passes (use -> ~0.99) but moves 32 bytes
    // per sample through HBM (store + ordered re-read), worth ~480
    // lane-instructions of a chip that issues ~6e13 of them per second against
    // ~4 TB/s.  Take the queue when the recovered work per sample is clearly
    // (3x) more than that.
    if (mode < 0) {
        uint32_t cand[3];
        int n_cand = 0;
        const bool scan_forced = (p->flags & (PTGPU_FLAG_SCAN_PLAIN | PTGPU_FLAG_SCAN_SCREENED |
                                              PTGPU_FLAG_SCAN_BVH)) != 0;
        const uint32_t n_obj = a.scene.n_obj;
        const bool has_bvh = a.scene.bvh_nodes != nullptr;
        if (scan_forced || n_obj < 4u) {
            cand[n_cand++] = a.screened;
        } else if (n_obj < 64u || !has_bvh) {
            cand[n_cand++] = 0u;
            cand[n_cand++] = 1u;
        } else if (n_obj < 512u) {
            cand[n_cand++] = 1u;
            cand[n_cand++] = 2u;
        } else { // (tools/bvh_scaling.py: 1.3x at 250 objects, 2.8x at 1000)
            cand[n_cand++] = 2u;
        }
        // 1 + 2 * 3 samples per candidate: 14 in all at most; can_measure
        // guarantees total_spp >= 16
        const uint32_t per_launch = 3u;
        const int rounds = 2;
        struct Tally {
            double ms = 0.0, rays = 0.0, samples = 0.0, lane_passes = 0.0;
        } tally[3];
        hipEvent_t ev[2] = {nullptr, nullptr};
        struct DropEvents { // (in place before the first create: neither can leak)
            hipEvent_t *ev;
            ~DropEvents() {
                if (ev[0] != nullptr)
                    (void)hipEventDestroy(ev[0]);
                if (ev[1] != nullptr)
                    (void)hipEventDestroy(ev[1]);
            }
        } drop_events{ev};
        PT_TRY_HIP(hipEventCreate(&ev[0]));
        PT_TRY_HIP(hipEventCreate(&ev[1]));
        if (accum_is_zero) {
            PT_TRY_HIP(hipMemsetAsync(accum, 0, (size_t)npix * 16u, stream));
            accum_is_zero = false;
        }
        auto pilot = [&](uint32_t scan, uint32_t count, Tally *t) -> int {
            pt::Counters before, after;
            PT_TRY_HIP(hipStreamSynchronize(stream));
            int prc = read_counters(a.counters, &before);
            if (prc != PTGPU_OK)
                return prc;
            PT_TRY_HIP(hipEventRecord(ev[0], stream));
            if (scan == 2u) {
                const uint32_t target = done_spp + count;
                while (done_spp < target) {
                    prc = queue_pass(2u, target - done_spp);
                    if (prc != PTGPU_OK)
                        return prc;
                }
            } else {
                pt::LaunchArgs q = a;
                q.screened = scan;
                q.spp = count;
                q.sample_offset = a.sample_offset + done_spp;
                hipError_t e = L.megakernel(q, owner_block, tree, stream);
                if (e != hipSuccess)
                    return fail_hip(e, "megakernel launch (pilot)");
                count_launch();
                timed.add_bytes(32ull * npix);
                done_spp += count;
            }
            PT_TRY_HIP(hipEventRecord(ev[1], stream));
            PT_TRY_HIP(hipStreamSynchronize(stream));
            prc = read_counters(a.counters, &after);
            if (prc != PTGPU_OK)
                return prc;
            if (t != nullptr) {
                float ms = 0.0f;
                PT_TRY_HIP(hipEventElapsedTime(&ms, ev[0], ev[1]));
                t->ms += ms;
                t->rays += (double)(after.rays - before.rays);
                t->samples += (double)(after.samples - before.samples);
                t->lane_passes += 64.0 * (double)(after.wave_iters - before.wave_iters);
            }
            return PTGPU_OK;
        };
        for (int i = 0; i < n_cand; ++i) { // warm-up, still real samples
            rc = pilot(cand[i], 1u, nullptr);
            if (rc != PTGPU_OK)
                return rc;
        }
        for (int k = 0; k < n_cand * rounds; ++k) {
            rc = pilot(cand[k % n_cand], per_launch, &tally[k % n_cand]);
            if (rc != PTGPU_OK)
                return rc;
        }
        int best = 0;
        for (int i = 1; i < n_cand; ++i)
            if (tally[i].ms < tally[best].ms)
                best = i;
        a.screened = cand[best];
        if (!scan_forced)
            s->scan_choice.store((int)cand[best]);
        const Tally &t = tally[best];
        const double rays_per_sample = t.samples > 0 ? t.rays / t.samples : 1.0;
        s->path_rays_milli.store((int)std::min(1e6, rays_per_sample * 1e3 + 0.5));
        if (a.screened == 2u) {
            mode = kSampleQueue; // resumable walks exist in queue form only
        } else {
            const double util = t.lane_passes > 0 && t.rays > 0 ? t.rays / t.lane_passes : 1.0;
            const double instr_per_ray = (a.screened == 1u ? 16.0 : 80.0) * (double)n_obj + 400.0;
            const double recovered = (1.0 / util - 1.0 / 0.99) * rays_per_sample * instr_per_ray;
            // (3x, not 1x: 3-sample pilot launches under-state the lane use a
            // full-length launch reaches, because their tail weighs more)
            mode = recovered > 3.0 * 480.0 ? kSampleQueue : kPixelOwner;
        }
        s->queue_mode.store(mode);
        s->last_mode.store(mode);
        remember_choice(s);
        choose_lanes(false); // for the rest of this frame
    }

    if (mode == kPixelOwner || done_spp >= total_spp) {
        // (done_spp == total_spp: the pilots were the whole frame)
        if (done_spp < total_spp) {
            if (accum_is_zero) {
                PT_TRY_HIP(hipMemsetAsync(accum, 0, (size_t)npix * 16u, stream));
                accum_is_zero = false;
            }
            pt::LaunchArgs q = a;
            q.spp = total_spp - done_spp;
            q.sample_offset = a.sample_offset + done_spp;
            hipError_t e = L.megakernel(q, owner_block, tree, stream);
            if (e != hipSuccess)
                return fail_hip(e, "megakernel launch (accumulate)");
            count_launch();
            timed.add_bytes(32ull * npix);
        }
        if (d_frame != nullptr) { // resolve the scratch accumulator
            pt::LaunchArgs q = a;
            q.spp = 0;
            q.first_pass = 0;
            q.frame = d_frame;
            hipError_t e = L.sum(q, stream);
            if (e != hipSuccess)
                return fail_hip(e, "resolve launch");
            timed.add_bytes(19ull * npix);
        }
        return PTGPU_OK;
    }

    // ---- sample-queue passes -----------------------------------------------------
    while (done_spp < total_spp) {
        rc = queue_pass(a.screened, total_spp - done_spp);
        if (rc != PTGPU_OK)
            return rc;
    }
    return PTGPU_OK;
}

} // namespace

int ptgpu_render_device(const ptgpu_scene *s, uint8_t *d_frame, size_t framew,
                        size_t frameh, const ptgpu_params *p,
                        void *hip_stream) {
    if (d_frame == nullptr)
        return fail(PTGPU_ERR_INVALID, "d_frame is NULL");
    return render_common(s, d_frame, nullptr, framew, frameh, p,
                         (hipStream_t)hip_stream, false);
}

int ptgpu_accumulate_device(const ptgpu_scene *s, float *d_accum,
                            size_t framew, size_t frameh,
                            const ptgpu_params *p, void *hip_stream) {
    if (d_accum == nullptr)
        return fail(PTGPU_ERR_INVALID, "d_accum is NULL");
    if ((reinterpret_cast<uintptr_t>(d_accum) & 15u) != 0)
        return fail(PTGPU_ERR_INVALID, "d_accum must be 16-byte aligned");
    return render_common(s, nullptr, d_accum, framew, frameh, p,
                         (hipStream_t)hip_stream, false);
}

int ptgpu_scene_tune(const ptgpu_scene *s, size_t framew, size_t frameh, const ptgpu_params *p) {
    if (s == nullptr)
        return fail(PTGPU_ERR_INVALID, "scene is NULL");
    ptgpu_params defaults;
    if (p == nullptr) {
        ptgpu_params_default(&defaults);
        defaults.spp = 64;
        p = &defaults;
    }
    if (framew == 0 || frameh == 0 || framew * frameh > 0xFFFFFFFFull)
        return fail(PTGPU_ERR_INVALID, "frame size out of range");
    uint8_t *scratch = nullptr;
    PT_TRY_HIP(hipMalloc(&scratch, 3u * framew * frameh));
    ptgpu_params q = *p;
    q.flags &= ~(uint32_t)PTGPU_FLAG_COMPACT_ROWS;
    int rc = render_common(s, scratch, nullptr, framew, frameh, &q, nullptr, true);
    const hipError_t e = hipDeviceSynchronize();
    (void)hipFree(scratch);
    if (rc == PTGPU_OK && e != hipSuccess)
        rc = fail_hip(e, "scene tune");
    return rc;
}

int ptgpu_resolve_device(const float *d_accum, uint8_t *d_frame,
                         size_t npixels, void *hip_stream) {
    if (d_accum == nullptr || d_frame == nullptr)
        return fail(PTGPU_ERR_INVALID, "NULL buffer");
    hipError_t e = pt::launch_resolve(d_accum, d_frame, (uint64_t)npixels,
                                      (hipStream_t)hip_stream);
    if (e != hipSuccess)
        return fail_hip(e, "resolve launch");
    return PTGPU_OK;
}

namespace {

// Global row of local row i of a call (same map as the kernel's).
uint32_t global_row(const pt::LaunchArgs &a, uint32_t i) {
    if (a.tile_rows == 0)
        return a.row_begin + i;
    const uint32_t tile = i / a.tile_rows;
    return a.row_begin + tile * a.tile_rows * a.tile_stride + (i % a.tile_rows);
}

// Device-side row buffer of the host-frame entry points, one per device.
struct RowBuffer {
    std::mutex lock;
    uint8_t *rows = nullptr;
    size_t bytes = 0;
};
RowBuffer g_row_buffers[kMaxDevices];

// `dest_compact`: `frame` is a shard buffer (local row i at row i) instead of
// a whole frame (local row i at its global row).
int render_frame_with(const ptgpu_scene *s, uint8_t *frame, size_t framew,
                      size_t frameh, const ptgpu_params *p, bool dest_compact = false) {
    ptgpu_params q = *p;
    q.flags |= PTGPU_FLAG_COMPACT_ROWS;
    pt::LaunchArgs a; // only for the row map and sizes
    uint32_t block = 0;
    bool tree = false;
    int rc = prepare_launch(s, framew, frameh, &q, &a, &block, &tree);
    if (rc != PTGPU_OK)
        return rc;
    const size_t row_bytes = 3u * framew;
    const size_t bytes = row_bytes * a.num_rows;
    // The rows are rendered into a device buffer that is kept from call to
    // call (a window loop calls this every frame; hipMalloc + hipFree were a
    // fifth of the reference scene's 0.9 ms frame).  A second host thread
    // rendering on the same device at the same time gets a buffer of its own.
    int dev = 0;
    rc = current_device(&dev);
    if (rc != PTGPU_OK)
        return rc;
    RowBuffer &keep = g_row_buffers[dev];
    std::unique_lock<std::mutex> mine(keep.lock, std::try_to_lock);
    uint8_t *d_rows = nullptr;
    if (mine.owns_lock()) {
        if (keep.bytes < bytes) {
            if (keep.rows != nullptr)
                (void)hipFree(keep.rows);
            keep.rows = nullptr;
            keep.bytes = 0;
            PT_TRY_HIP(hipMalloc(&keep.rows, bytes));
            keep.bytes = bytes;
        }
        d_rows = keep.rows;
    } else {
        PT_TRY_HIP(hipMalloc(&d_rows, bytes));
    }
    rc = render_common(s, d_rows, nullptr, framew, frameh, &q, nullptr, true);
    hipError_t e = hipSuccess;
    if (rc == PTGPU_OK)
        e = hipDeviceSynchronize();
    // copy back run by run (a run = consecutive local rows that are also
    // consecutive global rows)
    uint32_t i = 0;
    while (rc == PTGPU_OK && e == hipSuccess && i < a.num_rows) {
        uint32_t j = i + 1;
        while (j < a.num_rows && global_row(a, j) == global_row(a, j - 1) + 1)
            ++j;
        e = hipMemcpy(frame + row_bytes * (dest_compact ? i : global_row(a, i)),
                      d_rows + row_bytes * i, row_bytes * (j - i),
                      hipMemcpyDeviceToHost);
        i = j;
    }
    if (!mine.owns_lock())
        (void)hipFree(d_rows);
    if (rc != PTGPU_OK)
        return rc;
    if (e != hipSuccess)
        return fail_hip(e, "render_frame");
    return PTGPU_OK;
}

} // namespace

int ptgpu_render_frame(const scene *host_scene, uint8_t *frame, size_t framew,
                       size_t frameh, const ptgpu_params *p) {
    if (frame == nullptr)
        return fail(PTGPU_ERR_INVALID, "frame is NULL");
    ptgpu_params defaults;
    if (p == nullptr) {
        ptgpu_params_default(&defaults);
        p = &defaults;
    }
    ptgpu_scene *s = nullptr;
    int rc = ptgpu_scene_upload(host_scene, &s);
    if (rc != PTGPU_OK)
        return rc;
    rc = render_frame_with(s, frame, framew, frameh, p);
    ptgpu_scene_free(s);
    return rc;
}

// ---------------------------------------------------------------------------
// one process, several GPUs, host frame
// ---------------------------------------------------------------------------
int ptgpu_render_frame_multi(const scene *host_scene, uint8_t *frame,
                             size_t framew, size_t frameh,
                             const ptgpu_params *p, const int *devices,
                             int n_devices, uint32_t tile_rows) {
    if (frame == nullptr || devices == nullptr || n_devices <= 0 || n_devices > kMaxDevices)
        return fail(PTGPU_ERR_INVALID, "bad frame or device list");
    ptgpu_params defaults;
    if (p == nullptr) {
        ptgpu_params_default(&defaults);
        p = &defaults;
    }
    if (p->tile_rows != 0 || (p->flags & PTGPU_FLAG_COMPACT_ROWS))
        return fail(PTGPU_ERR_INVALID, "the multi-device call shards the rows itself");
    if (framew == 0 || frameh == 0 || p->row_begin >= frameh)
        return fail(PTGPU_ERR_INVALID, "frame size or row_begin out of range");
    const uint32_t first = p->row_begin;
    const uint32_t rows = p->num_rows ? p->num_rows : (uint32_t)frameh - first;
    if ((uint64_t)first + rows > frameh)
        return fail(PTGPU_ERR_INVALID, "row set reaches below the frame");
    const uint32_t T = tile_rows ? tile_rows : 8u;
    const uint32_t G = (uint32_t)n_devices;
    const uint32_t n_tiles = (rows + T - 1) / T;
    const size_t row_bytes = 3u * framew;

    int home = 0;
    PT_TRY_HIP(hipGetDevice(&home));
    struct Shard {
        ptgpu_scene *scene = nullptr;
        uint8_t *rows = nullptr;
        hipStream_t stream = nullptr;
        uint32_t num_rows = 0;
    };
    std::vector<Shard> shard(G);
    int rc = PTGPU_OK;
    // enqueue every device's tiles first, then collect: the devices overlap
    for (uint32_t r = 0; r < G && rc == PTGPU_OK; ++r) {
        uint32_t mine = 0;
        for (uint32_t t = r; t < n_tiles; t += G)
            mine += (t + 1) * T <= rows ? T : rows - t * T;
        shard[r].num_rows = mine;
        if (mine == 0)
            continue;
        hipError_t e = hipSetDevice(devices[r]);
        if (e == hipSuccess)
            e = hipStreamCreateWithFlags(&shard[r].stream, hipStreamNonBlocking);
        if (e == hipSuccess)
            e = hipMalloc(&shard[r].rows, row_bytes * mine);
        if (e != hipSuccess) {
            rc = fail_hip(e, "multi-device setup");
            break;
        }
        rc = ptgpu_scene_upload(host_scene, &shard[r].scene);
        if (rc != PTGPU_OK)
            break;
        ptgpu_params q = *p;
        q.row_begin = first + r * T;
        q.num_rows = mine;
        q.tile_rows = T;
        q.tile_stride = G;
        q.flags |= PTGPU_FLAG_COMPACT_ROWS;
        rc = render_common(shard[r].scene, shard[r].rows, nullptr, framew, frameh,
                           &q, shard[r].stream, true);
    }
    for (uint32_t r = 0; r < G; ++r) {
        if (shard[r].num_rows == 0)
            continue;
        (void)hipSetDevice(devices[r]);
        if (rc == PTGPU_OK && shard[r].stream) {
            hipError_t e = hipStreamSynchronize(shard[r].stream);
            uint32_t local = 0;
            for (uint32_t t = r; t < n_tiles && e == hipSuccess; t += G) {
                const uint32_t h = (t + 1) * T <= rows ? T : rows - t * T;
                e = hipMemcpy(frame + row_bytes * (first + t * T),
                              shard[r].rows + row_bytes * local, row_bytes * h,
                              hipMemcpyDeviceToHost);
                local += h;
            }
            if (e != hipSuccess)
                rc = fail_hip(e, "multi-device collect");
        }
        ptgpu_scene_free(shard[r].scene);
        if (shard[r].rows)
            (void)hipFree(shard[r].rows);
        if (shard[r].stream)
            (void)hipStreamDestroy(shard[r].stream);
    }
    (void)hipSetDevice(home);
    return rc;
}

// ---------------------------------------------------------------------------
// one process, several GPUs, frame assembled in HBM on devices[0]
// ---------------------------------------------------------------------------
} // extern "C"

namespace {

// librccl, resolved at first use: hosts that never render across devices do
// not load it, and a process in which PyTorch has already loaded its bundled
// copy (same SONAME) shares that one.
struct Rccl {
    void *lib = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string why; // non-empty: could not be loaded
};

const Rccl &rccl() {
    static const Rccl loaded = [] {
        Rccl r;
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (r.lib != nullptr)
                break;
        }
        if (r.lib == nullptr) {
            const char *e = dlerror();
            r.why = std::string("librccl could not be loaded: ") + (e ? e : "?");
            return r;
        }
        auto sym = [&](const char *name) {
            void *p = dlsym(r.lib, name);
            if (p == nullptr && r.why.empty())
                r.why = std::string("librccl lacks ") + name;
            return p;
        };
        r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(sym("ncclCommInitAll"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
        r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
        r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
        r.Send = reinterpret_cast<decltype(r.Send)>(sym("ncclSend"));
        r.Recv = reinterpret_cast<decltype(r.Recv)>(sym("ncclRecv"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
        return r;
    }();
    return loaded;
}

int fail_nccl(ncclResult_t e, const char *where) {
    const Rccl &r = rccl();
    return fail(PTGPU_ERR_HIP, std::string(where) + ": " +
                                   (r.GetErrorString ? r.GetErrorString(e) : "RCCL error"));
}

struct DeviceBuffer {
    uint8_t *ptr = nullptr;
    size_t bytes = 0;
};

// grows `b` on the CURRENT device
int ensure(DeviceBuffer *b, size_t need) {
    if (b->bytes >= need)
        return PTGPU_OK;
    if (b->ptr != nullptr)
        (void)hipFree(b->ptr);
    b->ptr = nullptr;
    b->bytes = 0;
    PT_TRY_HIP(hipMalloc(&b->ptr, need));
    b->bytes = need;
    return PTGPU_OK;
}

} // namespace

struct ptgpu_multi {
    std::vector<int> devices;
    std::vector<hipStream_t> streams;
    std::vector<hipEvent_t> rendered;    // shard r complete (peer-copy transport)
    std::vector<ncclComm_t> comms;       // empty with the peer-copy transport
    std::vector<ptgpu_scene *> scenes;
    std::vector<DeviceBuffer> shards;    // on device r (r >= 1)
    DeviceBuffer stage;                  // on devices[0]: every shard, back to back
    bool use_rccl = false;
    // RCCL ranks = the DISTINCT devices of the list, in order of first
    // appearance (rank 0 = devices[0]).  With all ordinals different (the
    // normal case) shard r is rank r; PTGPU_MULTI_RCCL_LOOPBACK lets ordinals
    // repeat, several shards then share a rank and its first shard's stream.
    std::vector<int> rank_of;            // per shard
    std::vector<int> lead_shard;         // per rank: its first shard (whose stream carries the rank's sends)
    std::mutex lock;
};

extern "C" {

void ptgpu_multi_destroy(ptgpu_multi *m) {
    if (m == nullptr)
        return;
    int home = 0;
    (void)hipGetDevice(&home);
    for (size_t r = 0; r < m->devices.size(); ++r) {
        (void)hipSetDevice(m->devices[r]);
        if (r < m->streams.size() && m->streams[r])
            (void)hipStreamSynchronize(m->streams[r]);
        if (r < m->comms.size() && m->comms[r] && rccl().CommDestroy)
            (void)rccl().CommDestroy(m->comms[r]);
        if (r < m->scenes.size())
            ptgpu_scene_free(m->scenes[r]);
        if (r < m->shards.size() && m->shards[r].ptr)
            (void)hipFree(m->shards[r].ptr);
        if (r == 0 && m->stage.ptr)
            (void)hipFree(m->stage.ptr);
        if (r < m->rendered.size() && m->rendered[r])
            (void)hipEventDestroy(m->rendered[r]);
        if (r < m->streams.size() && m->streams[r])
            (void)hipStreamDestroy(m->streams[r]);
    }
    (void)hipSetDevice(home);
    delete m;
}

int ptgpu_multi_create(const int *devices, int n_devices, uint32_t flags, ptgpu_multi **out) {
    if (out == nullptr)
        return fail(PTGPU_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (devices == nullptr || n_devices <= 0 || n_devices > (int)pt::kMaxShards)
        return fail(PTGPU_ERR_INVALID, "bad device list");
    if (flags & ~(PTGPU_MULTI_PEER_COPY | PTGPU_MULTI_RCCL_LOOPBACK))
        return fail(PTGPU_ERR_INVALID, "unknown flag");
    if ((flags & PTGPU_MULTI_PEER_COPY) && (flags & PTGPU_MULTI_RCCL_LOOPBACK))
        return fail(PTGPU_ERR_INVALID, "one transport at a time");
    const int have = ptgpu_device_count();
    if (have < 0)
        return have;
    for (int i = 0; i < n_devices; ++i)
        if (devices[i] < 0 || devices[i] >= have)
            return fail(PTGPU_ERR_INVALID, "device ordinal out of range");
    const bool peer_copy = (flags & PTGPU_MULTI_PEER_COPY) != 0;
    const bool loopback = (flags & PTGPU_MULTI_RCCL_LOOPBACK) != 0;
    if (!peer_copy) {
        // a communicator spans distinct devices, and a gather needs a peer
        std::vector<int> sorted(devices, devices + n_devices);
        std::sort(sorted.begin(), sorted.end());
        const bool distinct = std::adjacent_find(sorted.begin(), sorted.end()) == sorted.end();
        if (!loopback && (n_devices < 2 || !distinct))
            return fail(PTGPU_ERR_LIMIT, "the RCCL gather needs at least two distinct devices "
                                         "(PTGPU_MULTI_PEER_COPY rehearses on fewer)");
        if (loopback && n_devices < 2)
            return fail(PTGPU_ERR_LIMIT, "at least two shards, or nothing travels");
        if (!rccl().why.empty())
            return fail(PTGPU_ERR_NO_DEVICE, rccl().why);
    }
    ptgpu_multi *m = new (std::nothrow) ptgpu_multi();
    if (m == nullptr)
        return fail(PTGPU_ERR_NOMEM, "out of host memory");
    m->devices.assign(devices, devices + n_devices);
    m->streams.assign(n_devices, nullptr);
    m->rendered.assign(n_devices, nullptr);
    m->scenes.assign(n_devices, nullptr);
    m->shards.assign(n_devices, DeviceBuffer());
    m->use_rccl = !peer_copy;
    std::vector<int> rank_devices;
    m->rank_of.assign(n_devices, 0);
    for (int r = 0; r < n_devices; ++r) {
        size_t k = 0;
        while (k < rank_devices.size() && rank_devices[k] != devices[r])
            ++k;
        if (k == rank_devices.size()) {
            rank_devices.push_back(devices[r]);
            m->lead_shard.push_back(r);
        }
        m->rank_of[r] = (int)k;
    }
    int home = 0;
    hipError_t e = hipGetDevice(&home);
    for (int r = 0; r < n_devices && e == hipSuccess; ++r) {
        e = hipSetDevice(devices[r]);
        if (e == hipSuccess)
            e = hipStreamCreateWithFlags(&m->streams[r], hipStreamNonBlocking);
        if (e == hipSuccess)
            e = hipEventCreateWithFlags(&m->rendered[r], hipEventDisableTiming);
    }
    int rc = e == hipSuccess ? PTGPU_OK : fail_hip(e, "multi-device setup");
    if (rc == PTGPU_OK && m->use_rccl) {
        m->comms.assign(rank_devices.size(), nullptr);
        const ncclResult_t ne = rccl().CommInitAll(m->comms.data(), (int)rank_devices.size(),
                                                   rank_devices.data());
        if (ne != ncclSuccess)
            rc = fail_nccl(ne, "ncclCommInitAll");
    }
    (void)hipSetDevice(home);
    if (rc != PTGPU_OK) {
        const std::string keep = g_last_error;
        ptgpu_multi_destroy(m);
        g_last_error = keep;
        return rc;
    }
    *out = m;
    return PTGPU_OK;
}

const char *ptgpu_multi_transport(const ptgpu_multi *m) {
    return m == nullptr ? "" : (m->use_rccl ? "rccl" : "peer-copy");
}

int ptgpu_multi_upload(ptgpu_multi *m, const scene *host_scene) {
    if (m == nullptr || host_scene == nullptr)
        return fail(PTGPU_ERR_INVALID, "NULL argument");
    std::lock_guard<std::mutex> guard(m->lock);
    int home = 0;
    PT_TRY_HIP(hipGetDevice(&home));
    int rc = PTGPU_OK;
    for (size_t r = 0; r < m->devices.size() && rc == PTGPU_OK; ++r) {
        hipError_t e = hipSetDevice(m->devices[r]);
        if (e == hipSuccess)
            e = hipStreamSynchronize(m->streams[r]); // the old copy may still be in use
        if (e != hipSuccess) {
            rc = fail_hip(e, "multi-device upload");
            break;
        }
        ptgpu_scene_free(m->scenes[r]);
        m->scenes[r] = nullptr;
        rc = ptgpu_scene_upload(host_scene, &m->scenes[r]);
    }
    (void)hipSetDevice(home);
    return rc;
}

int ptgpu_multi_render(ptgpu_multi *m, uint8_t *d_frame, size_t framew, size_t frameh,
                       const ptgpu_params *p, uint32_t tile_rows) {
    if (m == nullptr || d_frame == nullptr)
        return fail(PTGPU_ERR_INVALID, "NULL argument");
    ptgpu_params defaults;
    if (p == nullptr) {
        ptgpu_params_default(&defaults);
        p = &defaults;
    }
    if (p->tile_rows != 0 || (p->flags & PTGPU_FLAG_COMPACT_ROWS))
        return fail(PTGPU_ERR_INVALID, "the multi-device call shards the rows itself");
    if (framew == 0 || frameh == 0 || p->row_begin >= frameh)
        return fail(PTGPU_ERR_INVALID, "frame size or row_begin out of range");
    const uint32_t first = p->row_begin;
    const uint32_t rows = p->num_rows ? p->num_rows : (uint32_t)frameh - first;
    if ((uint64_t)first + rows > frameh)
        return fail(PTGPU_ERR_INVALID, "row set reaches below the frame");
    std::lock_guard<std::mutex> guard(m->lock);
    const uint32_t G = (uint32_t)m->devices.size();
    for (uint32_t r = 0; r < G; ++r)
        if (m->scenes[r] == nullptr)
            return fail(PTGPU_ERR_INVALID, "no scene: call ptgpu_multi_upload first");
    const uint32_t T = tile_rows ? tile_rows : 8u;
    const uint32_t n_tiles = (rows + T - 1) / T;
    const size_t row_bytes = 3u * framew;
    if (row_bytes > 0xFFFFFFFFull)
        return fail(PTGPU_ERR_LIMIT, "row too long");

    // the deal: tile t -> shard t mod G; shard r's rows lie back to back at
    // offset[r] of the staging buffer on devices[0] (16-byte aligned)
    pt::ScatterArgs sa;
    std::memset(&sa, 0, sizeof sa);
    std::vector<uint32_t> shard_rows(G, 0);
    size_t total = 0;
    for (uint32_t r = 0; r < G; ++r) {
        for (uint32_t t = r; t < n_tiles; t += G)
            shard_rows[r] += (t + 1) * T <= rows ? T : rows - t * T;
        sa.shard_offset[r] = total;
        total += (row_bytes * shard_rows[r] + 15u) & ~(size_t)15u;
    }

    int home = 0;
    PT_TRY_HIP(hipGetDevice(&home));
    // Whichever way this function is left -- the early returns of a failed HIP
    // call below included -- every stream is drained first: kernels already
    // enqueued on other shards' streams write m->shards and m->stage, and the
    // contract is that the call returns when the frame is complete (or failed)
    // and the buffers are free for the next call.
    struct DrainAndRestore {
        ptgpu_multi *m;
        int home;
        ~DrainAndRestore() {
            for (size_t r = 0; r < m->devices.size(); ++r) {
                (void)hipSetDevice(m->devices[r]);
                (void)hipStreamSynchronize(m->streams[r]);
            }
            (void)hipSetDevice(home);
        }
    } drain_and_restore{m, home};

    PT_TRY_HIP(hipSetDevice(m->devices[0]));
    int rc = ensure(&m->stage, total ? total : 16);
    // ---- every device renders its tiles (the devices overlap) -----------------
    for (uint32_t r = 0; r < G && rc == PTGPU_OK; ++r) {
        if (shard_rows[r] == 0)
            continue;
        PT_TRY_HIP(hipSetDevice(m->devices[r]));
        uint8_t *dst = m->stage.ptr + sa.shard_offset[0];
        if (r != 0) {
            rc = ensure(&m->shards[r], row_bytes * shard_rows[r]);
            if (rc != PTGPU_OK)
                break;
            dst = m->shards[r].ptr;
        }
        ptgpu_params q = *p;
        q.row_begin = first + r * T;
        q.num_rows = shard_rows[r];
        q.tile_rows = T;
        q.tile_stride = G;
        q.flags |= PTGPU_FLAG_COMPACT_ROWS;
        rc = render_common(m->scenes[r], dst, nullptr, framew, frameh, &q, m->streams[r], true);
        if (rc == PTGPU_OK && r != 0)
            PT_TRY_HIP(hipEventRecord(m->rendered[r], m->streams[r]));
    }
    // ---- ONE exchange: shards 1..G-1 -> devices[0] ---------------------------------
    if (rc == PTGPU_OK && m->use_rccl) {
        const Rccl &R = rccl();
        ncclResult_t ne = R.GroupStart();
        for (uint32_t r = 1; r < G && ne == ncclSuccess; ++r) {
            if (shard_rows[r] == 0)
                continue;
            const size_t bytes = row_bytes * shard_rows[r];
            // (all ordinals different: rank r sends on its own stream.  Loopback:
            // the rank's lead stream sends, once this shard has been rendered)
            const int rank = m->rank_of[r];
            const hipStream_t send_stream = m->streams[m->lead_shard[rank]];
            if (m->lead_shard[rank] != (int)r) {
                const hipError_t we = hipStreamWaitEvent(send_stream, m->rendered[r], 0);
                if (we != hipSuccess) {
                    (void)R.GroupEnd();
                    return fail_hip(we, "hipStreamWaitEvent");
                }
            }
            ne = R.Send(m->shards[r].ptr, bytes, ncclUint8, 0, m->comms[rank], send_stream);
            if (ne == ncclSuccess)
                ne = R.Recv(m->stage.ptr + sa.shard_offset[r], bytes, ncclUint8, rank,
                            m->comms[0], m->streams[0]);
        }
        const ncclResult_t ge = R.GroupEnd();
        if (ne == ncclSuccess)
            ne = ge;
        if (ne != ncclSuccess)
            rc = fail_nccl(ne, "grouped ncclSend/ncclRecv");
    } else if (rc == PTGPU_OK) {
        PT_TRY_HIP(hipSetDevice(m->devices[0]));
        for (uint32_t r = 1; r < G; ++r) {
            if (shard_rows[r] == 0)
                continue;
            PT_TRY_HIP(hipStreamWaitEvent(m->streams[0], m->rendered[r], 0));
            PT_TRY_HIP(hipMemcpyPeerAsync(m->stage.ptr + sa.shard_offset[r], m->devices[0],
                                          m->shards[r].ptr, m->devices[r],
                                          row_bytes * shard_rows[r], m->streams[0]));
        }
    }
    // ---- rows into place, on devices[0] --------------------------------------------
    if (rc == PTGPU_OK) {
        PT_TRY_HIP(hipSetDevice(m->devices[0]));
        sa.stage = m->stage.ptr;
        sa.frame = d_frame;
        sa.row_bytes = (uint32_t)row_bytes;
        sa.row_begin = first;
        sa.num_rows = rows;
        sa.tile_rows = T;
        sa.shards = G;
        const hipError_t e = pt::launch_scatter_tiles(sa, m->streams[0]);
        if (e != hipSuccess)
            rc = fail_hip(e, "scatter launch");
    }
    // the frame is complete when devices[0]'s stream is; the others are waited
    // for too, so that their shard buffers may be reused by the next call
    for (uint32_t r = 0; r < G; ++r) {
        (void)hipSetDevice(m->devices[r]);
        const hipError_t e = hipStreamSynchronize(m->streams[r]);
        if (e != hipSuccess && rc == PTGPU_OK)
            rc = fail_hip(e, "multi-device render");
    }
    return rc;
}

int ptgpu_render_device_multi(const scene *host_scene, uint8_t *d_frame_on_device0,
                              size_t framew, size_t frameh, const ptgpu_params *p,
                              const int *devices, int n_devices, uint32_t tile_rows) {
    ptgpu_multi *m = nullptr;
    int rc = ptgpu_multi_create(devices, n_devices, 0u, &m);
    if (rc != PTGPU_OK)
        return rc;
    rc = ptgpu_multi_upload(m, host_scene);
    if (rc == PTGPU_OK)
        rc = ptgpu_multi_render(m, d_frame_on_device0, framew, frameh, p, tile_rows);
    const std::string keep = g_last_error;
    ptgpu_multi_destroy(m);
    g_last_error = keep;
    return rc;
}

// ---------------------------------------------------------------------------
// device memory helpers
// ---------------------------------------------------------------------------
int ptgpu_device_alloc(size_t bytes, void **out) {
    if (out == nullptr || bytes == 0)
        return fail(PTGPU_ERR_INVALID, "bad allocation request");
    *out = nullptr;
    void *p = nullptr;
    PT_TRY_HIP(hipMalloc(&p, bytes));
    hipError_t e = hipMemset(p, 0, bytes);
    if (e != hipSuccess) {
        (void)hipFree(p);
        return fail_hip(e, "hipMemset");
    }
    *out = p;
    return PTGPU_OK;
}

int ptgpu_device_free(void *ptr) {
    if (ptr != nullptr)
        PT_TRY_HIP(hipFree(ptr));
    return PTGPU_OK;
}

int ptgpu_device_upload(void *dst_device, const void *src_host, size_t bytes) {
    if (dst_device == nullptr || src_host == nullptr)
        return fail(PTGPU_ERR_INVALID, "NULL buffer");
    PT_TRY_HIP(hipMemcpy(dst_device, src_host, bytes, hipMemcpyHostToDevice));
    return PTGPU_OK;
}

int ptgpu_device_download(void *dst_host, const void *src_device, size_t bytes) {
    if (dst_host == nullptr || src_device == nullptr)
        return fail(PTGPU_ERR_INVALID, "NULL buffer");
    PT_TRY_HIP(hipMemcpy(dst_host, src_device, bytes, hipMemcpyDeviceToHost));
    return PTGPU_OK;
}

int ptgpu_device_synchronize(void) {
    PT_TRY_HIP(hipDeviceSynchronize());
    return PTGPU_OK;
}

// ---------------------------------------------------------------------------
// batched closest-hit queries
// ---------------------------------------------------------------------------
int ptgpu_trace_rays(const scene *host_scene, const float *origins,
                     const float *directions, size_t n, ray_hit *hits) {
    return ptgpu_trace_rays_scan(host_scene, origins, directions, n, hits, 0);
}

int ptgpu_trace_rays_scan(const scene *host_scene, const float *origins,
                          const float *directions, size_t n, ray_hit *hits,
                          uint32_t scan_flags) {
    if (n == 0)
        return PTGPU_OK;
    if (origins == nullptr || directions == nullptr || hits == nullptr)
        return fail(PTGPU_ERR_INVALID, "NULL ray buffer");
    static_assert(sizeof(ray_hit) == sizeof(pt::DeviceRayHit), "ray_hit size");
    ptgpu_scene *s = nullptr;
    int rc = ptgpu_scene_upload(host_scene, &s);
    if (rc != PTGPU_OK)
        return rc;
    // (the scan form is settled before anything is allocated: no buffer to
    // lose on this error path)
    int scan = 0;
    if (scan_flags & PTGPU_FLAG_SCAN_SCREENED)
        scan = 1;
    if (scan_flags & PTGPU_FLAG_SCAN_BVH) {
        if (s->desc.bvh_nodes == nullptr) {
            ptgpu_scene_free(s);
            return fail(PTGPU_ERR_INVALID, "scene has no hierarchy (fewer than 64 objects)");
        }
        scan = 2;
    }
    float *d_o = nullptr, *d_d = nullptr;
    pt::DeviceRayHit *d_h = nullptr;
    const size_t vec_bytes = 3u * sizeof(float) * n;
    hipError_t e = hipMalloc(&d_o, vec_bytes);
    if (e == hipSuccess) e = hipMalloc(&d_d, vec_bytes);
    if (e == hipSuccess) e = hipMalloc(&d_h, sizeof(pt::DeviceRayHit) * n);
    if (e == hipSuccess) e = hipMemcpy(d_o, origins, vec_bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_d, directions, vec_bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess)
        e = launchers_for(scan_flags)->trace(s->desc, d_o, d_d, n, d_h, scan, nullptr);
    if (e == hipSuccess)
        e = hipMemcpy(hits, d_h, sizeof(pt::DeviceRayHit) * n, hipMemcpyDeviceToHost);
    if (d_o) (void)hipFree(d_o);
    if (d_d) (void)hipFree(d_d);
    if (d_h) (void)hipFree(d_h);
    ptgpu_scene_free(s);
    if (e != hipSuccess)
        return fail_hip(e, "trace_rays");
    return PTGPU_OK;
}

int ptgpu_trace_rays_device(const ptgpu_scene *s, const float *d_origins,
                            const float *d_directions, size_t n, ray_hit *d_hits,
                            uint32_t scan_flags, void *hip_stream) {
    if (s == nullptr)
        return fail(PTGPU_ERR_INVALID, "scene is NULL");
    if (n == 0)
        return PTGPU_OK;
    if (d_origins == nullptr || d_directions == nullptr || d_hits == nullptr)
        return fail(PTGPU_ERR_INVALID, "NULL ray buffer");
    if (n > 0xFFFFFFFFull * 256ull)
        return fail(PTGPU_ERR_LIMIT, "too many rays for one call");
    int scan = 0;
    if (scan_flags & PTGPU_FLAG_SCAN_SCREENED)
        scan = 1;
    if (scan_flags & PTGPU_FLAG_SCAN_BVH) {
        if (s->desc.bvh_nodes == nullptr)
            return fail(PTGPU_ERR_INVALID, "scene has no hierarchy (fewer than 64 objects)");
        scan = 2;
    }
    hipError_t e = launchers_for(scan_flags)->trace(s->desc, d_origins, d_directions, n,
                                                   reinterpret_cast<pt::DeviceRayHit *>(d_hits),
                                                   scan, (hipStream_t)hip_stream);
    if (e != hipSuccess)
        return fail_hip(e, "trace_rays_device");
    return PTGPU_OK;
}

// ---------------------------------------------------------------------------
// bookkeeping
// ---------------------------------------------------------------------------
int ptgpu_stats_reset(void) {
    {
        int d = 0;
        if (current_device(&d) == PTGPU_OK) {
            (void)hipDeviceSynchronize();
            g_timing[d].harvest();
            std::lock_guard<std::mutex> guard(g_timing[d].lock);
            g_timing[d].ms = 0.0;
            g_timing[d].bytes = 0;
        }
    }
    pt::Counters *c = nullptr;
    int rc = counters_for_current_device(&c);
    if (rc != PTGPU_OK)
        return rc;
    PT_TRY_HIP(hipDeviceSynchronize());
    PT_TRY_HIP(hipMemset(c, 0, sizeof(pt::Counters) * pt::kCounterShards));
    int dev = 0;
    if (current_device(&dev) == PTGPU_OK)
        g_launches[dev].store(0);
    return PTGPU_OK;
}

int ptgpu_stats_read(ptgpu_stats *out) {
    if (out == nullptr)
        return fail(PTGPU_ERR_INVALID, "out is NULL");
    pt::Counters *c = nullptr;
    int rc = counters_for_current_device(&c);
    if (rc != PTGPU_OK)
        return rc;
    pt::Counters host;
    PT_TRY_HIP(hipDeviceSynchronize());
    rc = read_counters(c, &host);
    if (rc != PTGPU_OK)
        return rc;
    out->rays = host.rays;
    out->samples = host.samples;
    out->wave_iters = host.wave_iters;
    out->scans = host.scans;
    int dev = 0;
    out->launches = (current_device(&dev) == PTGPU_OK) ? g_launches[dev].load() : 0;
    out->kernel_ms = 0.0;
    out->hbm_bytes = 0;
    if (current_device(&dev) == PTGPU_OK) {
        g_timing[dev].harvest(); // the device is idle: every pair has completed
        std::lock_guard<std::mutex> guard(g_timing[dev].lock);
        out->kernel_ms = g_timing[dev].ms;
        out->hbm_bytes = g_timing[dev].bytes;
    }
    return PTGPU_OK;
}

int ptgpu_release_scratch(void) {
    int dev = 0;
    int rc = current_device(&dev);
    if (rc != PTGPU_OK)
        return rc;
    DeviceScratch &sc = g_scratch[dev];
    RowBuffer &keep = g_row_buffers[dev];
    // One order everywhere: the row buffer's lock before the scratch lock
    // (render_frame_with holds the former for the whole call and render_common
    // takes the latter inside it); std::scoped_lock also avoids the deadlock by
    // construction.
    std::scoped_lock both(keep.lock, sc.lock);
    PT_TRY_HIP(hipDeviceSynchronize());
    if (keep.rows != nullptr)
        (void)hipFree(keep.rows);
    keep.rows = nullptr;
    keep.bytes = 0;
    if (sc.samples)
        (void)hipFree(sc.samples);
    if (sc.accum)
        (void)hipFree(sc.accum);
    if (sc.primary)
        (void)hipFree(sc.primary);
    sc.samples = nullptr;
    sc.accum = nullptr;
    sc.primary = nullptr;
    sc.sample_bytes = sc.accum_bytes = sc.primary_bytes = 0;
    sc.primary_view = PrimaryView{};
    return PTGPU_OK;
}

const char *ptgpu_last_error(void) { return g_last_error.c_str(); }

const char *ptgpu_version(void) { return "path_tracer_c_amd 0.1 (gfx950)"; }

// ---------------------------------------------------------------------------
// legacy entry points
// ---------------------------------------------------------------------------
namespace {

[[noreturn]] void die(const char *who) {
    std::fprintf(stderr,
                 "path_tracer (MI355X path): %s failed: %s\n"
                 "  this library has no CPU renderer to fall back to.\n",
                 who, g_last_error.c_str());
    std::abort();
}

// What a legacy call does when the device path fails.  The reference's
// functions return void and cannot report anything (src/renderer.c:92,
// src/threadpool.c:84-101), so the default is to stop loudly: a picture that
// silently stays black is worse than no picture.  PTGPU_ON_ERROR=zero (read
// per call) is the opt-out for a host program that must survive a transient
// HIP failure: the call writes nothing, says why on stderr -- once per
// process -- and returns; ptgpu_last_error() holds the reason.
bool legacy_failed(const char *who) {
    const char *policy = std::getenv("PTGPU_ON_ERROR");
    if (policy == nullptr || std::strcmp(policy, "zero") != 0)
        die(who);
    static std::atomic<bool> said{false};
    if (!said.exchange(true))
        std::fprintf(stderr,
                     "path_tracer (MI355X path): %s failed: %s\n"
                     "  PTGPU_ON_ERROR=zero: the call wrote nothing; later failures are not reported here "
                     "(ptgpu_last_error()).\n",
                     who, g_last_error.c_str());
    return false;
}

// PTGPU_FRAME_SEED=auto: every frame rendered through the legacy calls gets the
// stream `PTGPU_SEED + number of frames before it`, the analogue of the
// reference's real-time loop, which draws a fresh random_seed per frame
// (src/main.c:276).  Without it a legacy frame of an unchanged scene is the
// same picture every time (PTGPU_SEED is all the legacy API can say).  A frame
// is one threadpool_wait_for_tasks that rendered rows, or, for render_task
// called directly, the calls up to the first row asked for a second time.
std::atomic<uint32_t> g_legacy_frames{0};
bool frame_seed_auto() {
    const char *e = std::getenv("PTGPU_FRAME_SEED");
    return e != nullptr && std::strcmp(e, "auto") == 0;
}

// One cached device scene for the legacy calls, revalidated by content hash
// so a caller may edit its scene between frames (the reference's RT loop
// does, src/main.c:260-263).
struct CachedScene {
    const scene *key = nullptr;
    uint64_t hash = 0;
    int device = -1;
    ptgpu_scene *dev = nullptr;
};
std::mutex g_cache_mutex;
CachedScene g_cache;

// Content hash of a byte range: four independent multiply-xorshift lanes over
// 8-byte words (a direct render_task call checks the whole scene per row, and a
// byte-at-a-time FNV of a 200 000-triangle scene was 10 ms per row).  Detects
// edits; not cryptographic.
uint64_t hash_bytes(const void *data, size_t n, uint64_t seed) {
    const unsigned char *p = static_cast<const unsigned char *>(data);
    constexpr uint64_t kMul = 0x9E3779B97F4A7C15ull;
    uint64_t h[4] = {seed ^ 0xCBF29CE484222325ull, seed + 0x100000001B3ull, ~seed, seed * kMul + 1u};
    while (n >= 32) {
        uint64_t w[4];
        std::memcpy(w, p, 32);
        for (int k = 0; k < 4; ++k) {
            h[k] = (h[k] ^ w[k]) * kMul;
            h[k] ^= h[k] >> 29;
        }
        p += 32;
        n -= 32;
    }
    uint64_t tail = (uint64_t)n;
    for (size_t i = 0; i < n; ++i)
        tail = (tail ^ p[i]) * 0x100000001B3ull;
    uint64_t out = tail;
    for (int k = 0; k < 4; ++k) {
        out = (out ^ h[k]) * kMul;
        out ^= out >> 32;
    }
    return out;
}

uint64_t scene_hash(const scene *w) {
    uint64_t h = hash_bytes(&w->num_objects, sizeof w->num_objects, 1);
    h = hash_bytes(&w->num_materials, sizeof w->num_materials, h);
    h = hash_bytes(&w->sky_color, sizeof w->sky_color, h);
    if (w->num_objects)
        h = hash_bytes(w->objects, w->num_objects * sizeof(shape), h);
    if (w->num_materials)
        h = hash_bytes(w->materials, w->num_materials * sizeof(shape_material), h);
    return h;
}

// Returns the cached device copy of `w` (content hash `h`), uploading if it
// changed; nullptr if that failed and the error policy lets the call return.
ptgpu_scene *legacy_scene(const scene *w, uint64_t h) {
    int dev = 0;
    if (current_device(&dev) != PTGPU_OK) {
        legacy_failed("hipGetDevice");
        return nullptr;
    }
    if (g_cache.dev != nullptr && g_cache.key == w && g_cache.hash == h &&
        g_cache.device == dev)
        return g_cache.dev;
    ptgpu_scene_free(g_cache.dev);
    g_cache = CachedScene();
    ptgpu_scene *s = nullptr;
    if (ptgpu_scene_upload(w, &s) != PTGPU_OK) {
        legacy_failed("scene upload");
        return nullptr;
    }
    g_cache.key = w;
    g_cache.hash = h;
    g_cache.device = dev;
    g_cache.dev = s;
    return s;
}

// Rows [row_begin, row_begin + num_rows) of the frame with the settings `p`
// (row fields are set here).  `hash`: the scene's content hash if the caller
// has it already.  False: failed, nothing written (PTGPU_ON_ERROR=zero).
bool legacy_render_rows(const scene *world, uint8_t *frame, size_t framew,
                        size_t frameh, uint32_t row_begin, uint32_t num_rows,
                        ptgpu_params p, bool dest_compact = false, const uint64_t *hash = nullptr) {
    std::lock_guard<std::mutex> guard(g_cache_mutex);
    p.row_begin = row_begin;
    p.num_rows = num_rows;
    p.tile_rows = 0;
    p.tile_stride = 1;
    uint32_t n_dev = 0;
    if (!dest_compact && env_u32("PTGPU_DEVICES", &n_dev) && n_dev > 1 && num_rows >= 2 * n_dev) {
        int have = ptgpu_device_count();
        if (have < 1)
            return legacy_failed("device count");
        if ((int)n_dev > kMaxDevices)
            n_dev = kMaxDevices;
        std::vector<int> devices(n_dev);
        for (uint32_t i = 0; i < n_dev; ++i)
            devices[i] = (int)(i % (uint32_t)have);
        if (ptgpu_render_frame_multi(world, frame, framew, frameh, &p, devices.data(),
                                     (int)n_dev, 0) != PTGPU_OK)
            return legacy_failed("multi-device render");
        return true;
    }
    ptgpu_scene *s = legacy_scene(world, hash != nullptr ? *hash : scene_hash(world));
    if (s == nullptr)
        return false;
    if (render_frame_with(s, frame, framew, frameh, &p, dest_compact) != PTGPU_OK)
        return legacy_failed("render");
    return true;
}

} // namespace

// ---------------------------------------------------------------------------
// render_task called DIRECTLY (not through the pool shim): a caller with its
// own loop or its own threads asks for one row per call, and a launch plus a
// device-to-host copy per row is 1080 serialised round trips per 1080p frame.
// The row must be in the caller's frame when the call returns (that is the
// function's contract, src/renderer.c:92-116), so work cannot be put off; it
// can be done AHEAD: when calls walk the rows in order, the next miss renders
// twice as many rows as the last one (1, 2, 4, ... up to the rest of the
// frame; downwards or upwards) into a host-side cache, and the calls that
// follow are a content check and a memcpy.  A row is a pure function of (scene
// content, settings, frame size, row), which is the cache's key -- a caller
// that edits its scene between calls gets fresh rows (the scene is hashed once
// per call, word-wise).  A 1080-row frame costs 11 launches instead of 1080;
// once a frame of a VIEW -- scene pointer, frame size, settings; not the
// content, not the frame's seed -- has been walked to its end the next one is
// fetched whole, so a caller that animates its scene or runs with
// PTGPU_FRAME_SEED=auto gets one launch per frame.  PTGPU_LEGACY_READAHEAD=0
// turns it off (one row per call, as before).
// ---------------------------------------------------------------------------
namespace {

struct RowCache {
    const scene *world = nullptr;
    uint64_t hash = 0;
    size_t framew = 0, frameh = 0;
    ptgpu_params params;          // settings the rows were rendered with (seed: the frame's)
    uint32_t row_begin = 0, num_rows = 0;
    std::vector<uint8_t> rows;    // num_rows * 3 * framew bytes
    uint32_t batch = 1;           // rows fetched by the last miss
    bool walked_to_the_end = false; // the last frame of this view was walked sequentially
    bool valid = false;
    // PTGPU_FRAME_SEED=auto: the frame being walked and the rows it has handed out
    bool frame_open = false;
    uint32_t frame_index = 0;
    std::vector<uint8_t> served;
};
RowCache g_row_cache;
std::mutex g_row_cache_mutex; // held for a whole direct call; taken BEFORE g_cache_mutex
                              // (which legacy_render_rows takes inside)

bool same_settings_but_seed(const ptgpu_params &a, const ptgpu_params &b) {
    return a.spp == b.spp && a.max_bounces == b.max_bounces &&
           __builtin_bit_cast(uint32_t, a.fov_rad) == __builtin_bit_cast(uint32_t, b.fov_rad) &&
           a.sample_offset == b.sample_offset &&
           (a.flags & (PTGPU_FLAG_FAST_MATH | PTGPU_FLAG_GLSL_SEMANTICS)) ==
               (b.flags & (PTGPU_FLAG_FAST_MATH | PTGPU_FLAG_GLSL_SEMANTICS));
}

bool readahead_enabled() {
    static const bool on = [] {
        uint32_t v = 1;
        env_u32("PTGPU_LEGACY_READAHEAD", &v);
        return v != 0;
    }();
    return on;
}

void legacy_one_row(const scene *world, uint8_t *frame, size_t framew, size_t frameh, uint32_t y) {
    const size_t row_bytes = 3u * framew;
    ptgpu_params now;
    ptgpu_params_from_env(&now);
    RowCache &c = g_row_cache;
    std::lock_guard<std::mutex> guard(g_row_cache_mutex);
    const bool same_view = c.world == world && c.framew == framew && c.frameh == frameh &&
                           same_settings_but_seed(c.params, now);
    if (frame_seed_auto()) {
        // a new frame begins with a new view, or with the first row that is
        // asked for a second time
        if (!same_view || !c.frame_open || c.served.size() != frameh || c.served[y]) {
            c.frame_index = g_legacy_frames.fetch_add(1);
            c.served.assign(frameh, 0);
            c.frame_open = true;
        }
        c.served[y] = 1;
        now.seed += c.frame_index;
    } else {
        c.frame_open = false;
    }
    if (!readahead_enabled()) {
        c.world = world;
        c.framew = framew;
        c.frameh = frameh;
        c.params = now;
        c.valid = false;
        legacy_render_rows(world, frame, framew, frameh, y, 1u, now);
        return;
    }
    const uint64_t h = scene_hash(world);
    const bool same_key = c.valid && same_view && c.hash == h && c.params.seed == now.seed;
    if (same_key && y >= c.row_begin && y < c.row_begin + c.num_rows) {
        std::memcpy(frame + row_bytes * y, c.rows.data() + row_bytes * (y - c.row_begin), row_bytes);
        return;
    }
    // a miss: how far ahead?  Twice the last batch when this row continues the
    // walk (just past the cached rows); the whole frame when the last frame of
    // this view was walked to its end and this one starts at the top
    // (the walk may also go UP the frame: the reference's own pool hands out
    // the newest task first, src/threadpool.c:28, i.e. the bottom row first)
    uint32_t ahead = 1;
    uint32_t begin = y;
    const uint32_t twice = c.batch < (1u << 30) ? 2u * c.batch : c.batch;
    if (same_key && y == c.row_begin + c.num_rows) { // continues downwards
        ahead = twice;
    } else if (same_key && y + 1u == c.row_begin) { // continues upwards
        ahead = twice < y + 1u ? twice : y + 1u;
        begin = y + 1u - ahead;
    } else if (same_view && c.walked_to_the_end && (y == 0 || y + 1u == frameh)) {
        ahead = (uint32_t)frameh; // the last frame was walked through: fetch this one whole
        begin = 0;
    }
    if ((uint64_t)begin + ahead > frameh)
        ahead = (uint32_t)(frameh - begin);
    c.valid = false;
    c.world = world;
    c.hash = h;
    c.framew = framew;
    c.frameh = frameh;
    c.params = now;
    c.row_begin = begin;
    c.num_rows = ahead;
    c.batch = ahead;
    c.walked_to_the_end = false;
    c.rows.resize(row_bytes * ahead);
    if (!legacy_render_rows(world, c.rows.data(), framew, frameh, begin, ahead, now,
                            /*dest_compact=*/true, &h))
        return; // (PTGPU_ON_ERROR=zero: nothing written, nothing cached)
    c.valid = true;
    std::memcpy(frame + row_bytes * y, c.rows.data() + row_bytes * (y - begin), row_bytes);
    if (begin == 0 || begin + ahead == frameh)
        c.walked_to_the_end = true;
}

} // namespace

uint32_t ptgpu_legacy_frames(void) { return g_legacy_frames.load(); }

void render_task(render_args *arg) {
    if (arg == nullptr || arg->frame == nullptr || arg->world == nullptr)
        return;
    if (arg->screen_y < 0 || (size_t)arg->screen_y >= arg->frameh)
        return;
    legacy_one_row(arg->world, arg->frame, arg->framew, arg->frameh, (uint32_t)arg->screen_y);
}

ray_hit trace_ray(const vec3s origin, const vec3s direction,
                  const scene *world) {
    ray_hit hit;
    std::memset(&hit, 0, sizeof hit);
    if (ptgpu_trace_rays(world, origin.raw, direction.raw, 1, &hit) != PTGPU_OK) {
        legacy_failed("trace_ray");
        std::memset(&hit, 0, sizeof hit);
        hit.distance = -1.0f; // (PTGPU_ON_ERROR=zero: the reference's "no hit", src/ray.c:6)
    }
    return hit;
}

// The pool shim (host/threadpool.c) hands over every row queued so far.
void ptgpu_pool_render_rows(const render_args *queued, size_t n) {
    // group by (frame, size, scene); render each group's rows as contiguous runs
    // (a row outside its frame is dropped by itself -- the reference would write
    // outside the buffer -- not together with the run of rows it sorts into)
    std::vector<const render_args *> rows;
    rows.reserve(n);
    for (size_t i = 0; i < n; i++) {
        const render_args &r = queued[i];
        if (r.frame != nullptr && r.world != nullptr && r.screen_y >= 0 &&
            (size_t)r.screen_y < r.frameh)
            rows.push_back(&r);
    }
    std::stable_sort(rows.begin(), rows.end(), [](const render_args *a, const render_args *b) {
        // (std::less: `<` on pointers into different objects is unspecified)
        if (a->frame != b->frame) return std::less<const void *>()(a->frame, b->frame);
        if (a->world != b->world) return std::less<const void *>()(a->world, b->world);
        if (a->framew != b->framew) return a->framew < b->framew;
        if (a->frameh != b->frameh) return a->frameh < b->frameh;
        return a->screen_y < b->screen_y;
    });
    if (rows.empty())
        return;
    ptgpu_params settings;
    ptgpu_params_from_env(&settings);
    if (frame_seed_auto()) // one wait = one frame (every frame buffer of this wait)
        settings.seed += g_legacy_frames.fetch_add(1);
    size_t i = 0;
    while (i < rows.size()) {
        const render_args &head = *rows[i];
        size_t j = i + 1;
        while (j < rows.size() && rows[j]->frame == head.frame && rows[j]->world == head.world &&
               rows[j]->framew == head.framew && rows[j]->frameh == head.frameh &&
               (rows[j]->screen_y == rows[j - 1]->screen_y ||
                rows[j]->screen_y == rows[j - 1]->screen_y + 1))
            ++j;
        const int y0 = head.screen_y, y1 = rows[j - 1]->screen_y;
        legacy_render_rows(head.world, head.frame, head.framew, head.frameh, (uint32_t)y0,
                           (uint32_t)(y1 - y0 + 1), settings);
        i = j;
    }
}

} // extern "C"
