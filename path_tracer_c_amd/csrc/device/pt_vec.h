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

namespace pt {

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

// x*x' + y*y' + z*z', summed left to right (glms_vec3_dot).
PT_HD float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

PT_HD V3 cross(V3 a, V3 b) {
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
#if defined(__HIP_DEVICE_COMPILE__)
    if (__builtin_expect(x >= 0x1.0p-96f, 1))
        return sqrt_core(x);
#endif
    return sqrtf(x);
}

PT_HD float rcp_rn(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t mag = f32_bits(x) & 0x7FFFFFFFu;
    if (__builtin_expect(mag - 0x20800000u <= 0x3E000000u, 1)) // [2^-62, 2^62]
        return rcp_core(x);
#endif
    return 1.0f / x;
}

// glms_vec3_normalize: zero vector below FLT_EPSILON, else a * (1/|a|).
PT_HD V3 normalize(V3 a) {
    const float len2 = dot(a, a);
#if defined(__HIP_DEVICE_COMPILE__)
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

// glms_vec3_lerp: from + (to - from) * t, not clamped.
PT_HD V3 lerp(V3 from, V3 to, float t) { return from + (to - from) * t; }

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
    return v - n * twice;
}

// glms_vec3_refract: false (and a zero vector) on total internal reflection.
PT_HD bool refract(V3 v, V3 n, float eta, V3 &out) {
    float ndi = dot(n, v);
    float eni = eta * ndi;
    float k = 1.0f - eta * eta + eni * eni;
    if (k < 0.0f) {
        out = splat(0.0f);
        return false;
    }
    float along = eni + sqrt_rn(k);
    out = v * eta - n * along;
    return true;
}

} // namespace pt
