// pt_vec.h -- binary32 3-vector arithmetic of the render path, for gfx950.
//
// The reference does its vector math through cglm's struct API (17 glms_*
// functions); cglm is not in the reference tree, so
// these are written from the published definitions.  Every function spells
// out its evaluation order, because the path's contract is that the GPU
// stores the same bytes a sequential IEEE-754 evaluation stores: the
// translation unit is built with -ffp-contract=off and with correctly
// rounded division and square root (hipcc's default,
// -fhip-fp32-correctly-rounded-divide-sqrt), and fp32 denormals stay enabled.
//
// __host__ __device__: the host half is used when a scene is uploaded
// (triangle edges/normals and emitted radiance are computed once on the
// host with the very same operations the per-ray code would use).
#pragma once

#include <hip/hip_runtime.h>

#include <cfloat>
#include <cmath>
#include <cstdint>

#define PT_HD __host__ __device__ __forceinline__

// Wave-level votes on a bool.  HIP's __ballot / __any / __all take an int: a
// predicate is first written out as 0 / 1 into a VGPR and compared again
// (v_cndmask + v_cmp per vote; the inner-node step of the walk votes once per
// step).  The builtin takes the condition as it is: one s_and of the
// comparison's own mask.
__device__ __forceinline__ unsigned long long pt_ballot(bool p) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_ballot_w64(p);
#else
    return p ? 1ull : 0ull; // (host pass of the same translation unit: never called)
#endif
}
__device__ __forceinline__ bool pt_any(bool p) { return pt_ballot(p) != 0ull; }
__device__ __forceinline__ bool pt_all(bool p) { return pt_ballot(!p) == 0ull; }

// The device code is compiled twice (Makefile): as namespace `pt` under the
// bit-exact contract described above, and -- with -DPT_FAST_MATH
// -DPT_NS=pt_fast -- as namespace `pt_fast`, the opt-in tolerance mode behind
// PTGPU_FLAG_FAST_MATH (include/ptgpu.h): fused multiply-adds, v_sqrt_f32 /
// v_rcp_f32 / v_rsq_f32 without their correction steps.  The fusions and approximations are WRITTEN OUT (madd / dot /
// cross / the builtins below) and the build keeps the exact build's compiler
// flags (-ffp-contract=off, IEEE `/` and sqrtf): left to the compiler
// (-ffp-contract=fast, approximate division) the same expression came out
// differently in different kernels -- a sample queue frame differed from the
// pixel-owner frame of the same scene (found by the mode's fuzz test) -- this
// way every kernel of the mode does the same arithmetic, as in the exact build.  Same algorithm, same random stream, same
// launch plumbing; results agree with the exact build to the tolerance
// tests/test_parity_gpu.py states, not to the bit.  The layout types
// (pt_layout.h, pt_launch.h) live in `pt` for both.
#ifndef PT_NS
#define PT_NS pt
#endif
// (the two halves of the mode can be switched off one by one when looking for
// the source of a difference: -DPT_EXACT_SPHERE, -DPT_EXACT_RECIP)
// (-DPT_FAST_MINIMAL: an experiment, never shipped -- the tolerance build's
// namespace and plumbing with the EXACT arithmetic everywhere except one fused
// multiply-add, the hit point of surface_at (pt_trace.h): how far does the
// smallest possible perturbation move configuration 5's frame?  ptgpu.h,
// PTGPU_FLAG_FAST_MATH, has the answer.)
#if defined(PT_FAST_MATH) && !defined(PT_FAST_MINIMAL)
#define PT_FUSE 1
#endif
#if defined(PT_FUSE) && !defined(PT_EXACT_SPHERE)
#define PT_FAST_SPHERE 1
#endif
#if defined(PT_FUSE) && !defined(PT_EXACT_RECIP)
#define PT_FAST_RECIP 1
#endif
#ifdef PT_FAST_MATH
#include "pt_layout.h"
namespace pt_fast {
using namespace pt;
}
#endif

namespace PT_NS {

struct V3 {
    float x, y, z;
};

PT_HD V3 v3(float x, float y, float z) { return V3{x, y, z}; }
PT_HD V3 splat(float s) { return V3{s, s, s}; }

PT_HD V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
PT_HD V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
PT_HD V3 operator*(V3 a, V3 b) { return V3{a.x * b.x, a.y * b.y, a.z * b.z}; }
PT_HD V3 operator*(V3 a, float s) { return V3{a.x * s, a.y * s, a.z * s}; }
PT_HD V3 operator-(V3 a) { return V3{-a.x, -a.y, -a.z}; }

// a * b + c: two roundings in the exact build (as the reference's separate
// operations round), one fused multiply-add in the tolerance build.
PT_HD float mul_add(float a, float b, float c) {
#if defined(PT_FUSE) && defined(__HIP_DEVICE_COMPILE__) && !defined(PT_X_MADD)
    return __builtin_fmaf(a, b, c);
#else
    return a * b + c;
#endif
}
PT_HD V3 madd(V3 a, float s, V3 b) { // a * s + b
    return V3{mul_add(a.x, s, b.x), mul_add(a.y, s, b.y), mul_add(a.z, s, b.z)};
}
PT_HD V3 madd(V3 a, V3 s, V3 b) { // a * s + b, component-wise
    return V3{mul_add(a.x, s.x, b.x), mul_add(a.y, s.y, b.y), mul_add(a.z, s.z, b.z)};
}

// x*x' + y*y' + z*z', summed left to right (glms_vec3_dot).
PT_HD float dot(V3 a, V3 b) {
#if defined(PT_FUSE) && defined(__HIP_DEVICE_COMPILE__) && !defined(PT_X_DOT)
    return __builtin_fmaf(a.z, b.z, __builtin_fmaf(a.y, b.y, a.x * b.x));
#else
    return a.x * b.x + a.y * b.y + a.z * b.z;
#endif
}

// The same sum with every product and sum rounded on its own in BOTH builds
// (the tolerance build needs it in two places: pt_trace.h ray_sphere,
// normalize_dir below).
PT_HD float dot_seq(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

PT_HD V3 cross(V3 a, V3 b) {
#if defined(PT_FUSE) && defined(__HIP_DEVICE_COMPILE__) && !defined(PT_X_DOT)
    return V3{__builtin_fmaf(a.y, b.z, -(a.z * b.y)), __builtin_fmaf(a.z, b.x, -(a.x * b.z)),
              __builtin_fmaf(a.x, b.y, -(a.y * b.x))};
#else
    return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z,
              a.x * b.y - a.y * b.x};
#endif
}

// cross product with every product and difference rounded on its own, in both builds
PT_HD V3 cross_seq(V3 a, V3 b) {
    return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z,
              a.x * b.y - a.y * b.x};
}

// ---- correctly rounded sqrt and reciprocal, cheaper than the generic ones ----
// hipcc's IEEE sqrtf is: range scaling + v_sqrt_f32 (<= 1 ulp) + a +-1 ulp
// correction from two fma residuals + special-value selects (17 VALU); its
// IEEE 1.0f/x is the 11-instruction div_scale/div_fmas/div_fixup sequence.
// On the ranges below the scaling and fix-ups are dead weight:
//   sqrt_rn(x), x >= 2^-96 (also +inf): v_sqrt_f32 + the same correction
//   rcp_rn(x),  2^-62 <= |x| <= 2^62  : v_rcp_f32 + ONE fma Newton step
// Both were compared with hipcc's sqrtf / 1.0f/x on ALL 2^32 bit patterns on an
// MI355X (tools/math_exhaustive.hip): zero mismatches inside those ranges.
// Outside them the generic code runs, so results are IEEE everywhere.
PT_HD uint32_t f32_bits(float f) { return __builtin_bit_cast(uint32_t, f); }
PT_HD float bits_f32(uint32_t u) { return __builtin_bit_cast(float, u); }

#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ float sqrt_core(float x) { // x >= 2^-96
    float s = __builtin_amdgcn_sqrtf(x);
    float below = bits_f32(f32_bits(s) - 1u), above = bits_f32(f32_bits(s) + 1u);
    float res_below = __builtin_fmaf(-below, s, x);
    float res_above = __builtin_fmaf(-above, s, x);
    s = (res_below <= 0.0f) ? below : s;
    s = (res_above > 0.0f) ? above : s;
    return s;
}
__device__ __forceinline__ float rcp_core(float x) { // 2^-62 <= |x| <= 2^62
    float y = __builtin_amdgcn_rcpf(x);
    float e = __builtin_fmaf(-x, y, 1.0f);
    return __builtin_fmaf(e, y, y);
}
#endif

PT_HD float sqrt_rn(float x) {
#if defined(__HIP_DEVICE_COMPILE__) && defined(PT_FAST_RECIP)
    return __builtin_amdgcn_sqrtf(x); // v_sqrt_f32, 1 ulp
#else
#if defined(__HIP_DEVICE_COMPILE__)
    if (__builtin_expect(x >= 0x1.0p-96f, 1))
        return sqrt_core(x);
#endif
    return sqrtf(x);
#endif
}

// correctly rounded 1/x in BOTH builds (the tolerance build needs it where a
// result must agree with the exact build's: pt_trace.h, ray_triangle)
PT_HD float rcp_ieee(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t mag = f32_bits(x) & 0x7FFFFFFFu;
    if (__builtin_expect(mag - 0x20800000u <= 0x3E000000u, 1)) // [2^-62, 2^62]
        return rcp_core(x);
#endif
    return 1.0f / x;
}

PT_HD float rcp_rn(float x) {
#if defined(__HIP_DEVICE_COMPILE__) && defined(PT_FAST_RECIP)
    return __builtin_amdgcn_rcpf(x); // v_rcp_f32, 1 ulp
#else
    return rcp_ieee(x);
#endif
}

// glms_vec3_normalize: zero vector below FLT_EPSILON, else a * (1/|a|).
PT_HD V3 normalize(V3 a) {
    const float len2 = dot(a, a);
#if defined(__HIP_DEVICE_COMPILE__) && defined(PT_FAST_RECIP) && !defined(PT_EXACT_NORMALIZE)
    // len < FLT_EPSILON <=> len2 < FLT_EPSILON^2 (up to rounding); v_rsq_f32
    if (!(len2 >= FLT_EPSILON * FLT_EPSILON))
        return (len2 != len2) ? splat(len2) : splat(0.0f);
    // One Newton step on top of v_rsq_f32.  The instruction alone is good to
    // 1 ulp but sits 0.085 ulp low on average (tools/rsq_bias.hip), and unit
    // vectors that are all a hair short made configuration 5's paths 0.8 %
    // longer and its image 0.15 of an 8-bit level darker than the exact
    // build's (tools/fast_math_probe3.py); three more instructions remove it.
    const float y = __builtin_amdgcn_rsqf(len2);
    const float e = __builtin_fmaf(-(len2 * y), y, 1.0f);
    return a * __builtin_fmaf(0.5f * y, e, y);
#elif defined(__HIP_DEVICE_COMPILE__)
    // one range test covers both fast paths: 2^-96 <= len2 <= 2^124 puts
    // len in [2^-48, 2^62]
    if (__builtin_expect(f32_bits(len2) - 0x0F800000u <= 0x6E000000u, 1)) {
        const float len = sqrt_core(len2);
        if (len < FLT_EPSILON)
            return splat(0.0f);
        return a * rcp_core(len);
    }
#endif
    float len = sqrtf(len2);
    if (len < FLT_EPSILON)
        return splat(0.0f);
    float inv = 1.0f / len;
    return a * inv;
}

// Normalisation of a RAY DIRECTION: the reference's chain -- correctly rounded
// sqrt, correctly rounded reciprocal, three products -- in BOTH builds.
// ray_sphere takes the direction for a unit vector, so a length error eps
// misplaces a hit by eps * distance; rays that have leaked into a radius-1000
// sphere (reflected below the surface; the reference lets them) travel chords
// of up to 2000 units, where 1e-7 * 2000 = 2e-4 is of the order of the 0.001
// origin offset and of that sphere's 6e-5 position rounding: whether such a
// ray stays trapped inside until the bounce limit or escapes depends on the
// SIGN of eps.  Measured on configuration 5 (tools/fast_math_probe3.py, 8
// bounces): all reflected directions scaled by 1 -/+ 1e-7: 3.42 / 3.33 rays per
// sample against 3.37, and the image 0.3 / 0.1 of an 8-bit level darker /
// brighter.  v_rsq_f32 (+ a Newton step) is as accurate as this chain but
// distributes eps differently, which made the tolerance build's paths 0.6 %
// longer and its picture 0.12 levels darker; with this chain at the three
// direction sites it is 0.01 % and 0.02 levels.  Normals keep the cheap form.
PT_HD V3 normalize_dir(V3 a) {
#if defined(__HIP_DEVICE_COMPILE__) && defined(PT_FAST_RECIP)
    const float len2 = dot_seq(a, a);
    if (__builtin_expect(f32_bits(len2) - 0x0F800000u <= 0x6E000000u, 1)) {
        const float len = sqrt_core(len2);
        if (len < FLT_EPSILON)
            return splat(0.0f);
        return a * rcp_core(len);
    }
    return normalize(a);
#else
    return normalize(a);
#endif
}

// glms_vec3_lerp: from + (to - from) * t, not clamped.
PT_HD V3 lerp(V3 from, V3 to, float t) {
#if defined(PT_FUSE) && defined(__HIP_DEVICE_COMPILE__)
    return madd(to - from, t, from);
#else
    return from + (to - from) * t;
#endif
}

// glm_clamp = min(max(v, lo), hi) built from `a > b ? a : b`; NaN -> lo.
PT_HD float clampf(float v, float lo, float hi) {
    float up = (v > lo) ? v : lo;
    return (up < hi) ? up : hi;
}
PT_HD V3 clamp01(V3 v) {
    return V3{clampf(v.x, 0.0f, 1.0f), clampf(v.y, 0.0f, 1.0f),
              clampf(v.z, 0.0f, 1.0f)};
}

// glms_vec3_reflect: v - n * (2 * dot(v, n)).
PT_HD V3 reflect(V3 v, V3 n) {
    float twice = 2.0f * dot(v, n);
#if defined(PT_FAST_MINIMAL) && PT_FAST_MINIMAL == 2 && defined(__HIP_DEVICE_COMPILE__)
    return V3{__builtin_fmaf(n.x, -twice, v.x), __builtin_fmaf(n.y, -twice, v.y),
              __builtin_fmaf(n.z, -twice, v.z)}; // (the experiment's one fused operation, variant 2)
#elif defined(PT_FUSE) && defined(__HIP_DEVICE_COMPILE__)
    return madd(n, -twice, v);
#else
    return v - n * twice;
#endif
}

// glms_vec3_refract: false (and a zero vector) on total internal reflection.
PT_HD bool refract(V3 v, V3 n, float eta, V3 &out) {
    float ndi = dot(n, v);
    float eni = eta * ndi;
#if defined(PT_FUSE) && defined(__HIP_DEVICE_COMPILE__)
    float k = __builtin_fmaf(eni, eni, __builtin_fmaf(-eta, eta, 1.0f));
#else
    float k = 1.0f - eta * eta + eni * eni;
#endif
    if (k < 0.0f) {
        out = splat(0.0f);
        return false;
    }
    float along = eni + sqrt_rn(k);
#if defined(PT_FUSE) && defined(__HIP_DEVICE_COMPILE__)
    out = madd(n, -along, v * eta);
#else
    out = v * eta - n * along;
#endif
    return true;
}

} // namespace PT_NS
