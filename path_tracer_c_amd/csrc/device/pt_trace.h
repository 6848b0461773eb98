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

namespace pt {

// A view of the scene records, in LDS or in global memory.
struct SceneRecords {
    const F4 *geom;
    const F4 *tri;
    const F4 *mat;
    const uint32_t *meta;
    uint32_t n_obj;
};

PT_HD V3 xyz(const F4 &r) { return V3{r.x, r.y, r.z}; }

// glms_ray_sphere for a unit direction; `r2` is radius*radius.
// Returns false when there is no root or both roots are behind the origin.
PT_HD bool ray_sphere(V3 o, V3 d, V3 c, float r2, float &t1, float &t2) {
    V3 dp = c - o;
    float ddp = dot(d, dp);
    float dpp = dot(dp, dp);
    dp = dp - d * ddp; // remedy term: component of dp orthogonal to d
    float dscr = r2 - dot(dp, dp);
    if (dscr < 0.0f)
        return false;
    dscr = sqrt_rn(dscr);
    float q = (ddp >= 0.0f) ? (ddp + dscr) : (ddp - dscr);
    float a = q;
    float b = (dpp - r2) / q;
    if (a > b) {
        float keep = a;
        a = b;
        b = keep;
    }
    t1 = a;
    t2 = b;
    return !(a < 0.0f && b < 0.0f);
}

// glms_ray_triangle (Moeller-Trumbore, eps 1e-6, two-sided) with the edges
// e1 = v1 - v0 and e2 = v2 - v0 taken from the uploaded record.
PT_HD bool ray_triangle(V3 o, V3 d, V3 v0, V3 e1, V3 e2, float &dist) {
    const float eps = 0.000001f;
    V3 p = cross(d, e2);
    float det = dot(e1, p);
    if (det > -eps && det < eps)
        return false;
    float inv_det = rcp_rn(det);
    V3 t = o - v0;
    float u = inv_det * dot(t, p);
    if (u < 0.0f || u > 1.0f)
        return false;
    V3 q = cross(t, e1);
    float v = inv_det * dot(d, q);
    if (v < 0.0f || u + v > 1.0f)
        return false;
    dist = inv_det * dot(e2, q);
    return dist > eps;
}

struct Hit {
    float distance; // < 0: miss
    int object;     // index into the scene's object array, -1 on a miss
};

// The scan of src/ray.c:12-65: keep a candidate unless
// (dist >= closest && closest > 0).
PT_HD Hit closest_hit(const SceneRecords &rec, V3 o, V3 d) {
    Hit h;
    h.distance = -1.0f;
    h.object = -1;
    for (uint32_t i = 0; i < rec.n_obj; ++i) {
        const F4 g = rec.geom[i];
        const uint32_t meta = rec.meta[i];
        float dist;
        bool found;
        if (!(meta & kMetaTriangle)) {
            float t1, t2;
            found = ray_sphere(o, d, xyz(g), g.w, t1, t2);
            dist = (t1 > 0.0f) ? t1 : t2; // src/ray.c:24
        } else {
            const uint32_t k = __builtin_bit_cast(uint32_t, g.w);
            const F4 e1 = rec.tri[3 * k + 0];
            const F4 e2 = rec.tri[3 * k + 1];
            found = ray_triangle(o, d, xyz(g), xyz(e1), xyz(e2), dist);
        }
        if (found && !(dist >= h.distance && h.distance > 0.0f)) {
            h.distance = dist;
            h.object = (int)i;
        }
    }
    return h;
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
    const F4 g = rec.geom[h.object];
    const uint32_t meta = rec.meta[h.object];
    s.material = meta & kMetaMaterialMask;
    s.point = d * h.distance + o;
    if (!(meta & kMetaTriangle)) {
        s.normal = normalize(s.point - xyz(g));
    } else {
        const uint32_t k = __builtin_bit_cast(uint32_t, g.w);
        V3 n = xyz(rec.tri[3 * k + 2]);
        if (dot(n, d) > 0.0f)
            n = -n;
        s.normal = n;
    }
    return s;
}

} // namespace pt
