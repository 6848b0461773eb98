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

__global__ void __launch_bounds__(256) pt_glsl_kernel(const LaunchArgs a) {
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
            const GHit hit = g_trace(a.scene, cur.origin, cur.direction);
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
    hipLaunchKernelGGL(pt_glsl_kernel, dim3(grid), dim3(block), 0, stream, a);
    return hipGetLastError();
}

} // namespace pt
