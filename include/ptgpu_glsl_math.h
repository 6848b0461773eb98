/*
 * ptgpu_glsl_math.h -- log() and cos() of the GLSL-semantics mode
 * (PTGPU_FLAG_GLSL_SEMANTICS), written as code so that the kernel
 * (csrc/device/pt_glsl.hip) and its checker (oracle/pt_glsl.c) evaluate the
 * shader's Box-Muller transform (shaders/rtx_frag.glsl:78-84) to the same bits.
 *
 * GLSL leaves the precision of its built-ins to the implementation, so any
 * faithful evaluation is "a" GLSL; libm's logf / cosf and the device's differ
 * in the last bit, and in a path tracer a last bit is, a few bounces later,
 * another path.  These two are the classic single-precision kernels (Cephes
 * logf / sinf / cosf: argument reduction + a short polynomial, 1-2 ulp) in
 * sequentially rounded binary32 operations only -- no fused multiply-add (both
 * sides are built with contraction off), no table, no library call -- hence
 * bit-identical on x86-64 and gfx950.  A specification like ptgpu_rng.h, not
 * an implementation either side could get away with changing alone.
 */
#ifndef PTGPU_GLSL_MATH_H
#define PTGPU_GLSL_MATH_H

#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define PTGPU_GM __host__ __device__ __forceinline__
#else
#define PTGPU_GM static inline
#endif

PTGPU_GM uint32_t ptgpu_gm_bits(float f) {
    uint32_t u;
    memcpy(&u, &f, sizeof u);
    return u;
}
PTGPU_GM float ptgpu_gm_float(uint32_t u) {
    float f;
    memcpy(&f, &u, sizeof f);
    return f;
}

/* natural logarithm of a normal float in (0, 1] (the stream's values are k/2^32,
 * k >= 1: never subnormal); 0 -> -infinity, as log(0) is in the shader */
PTGPU_GM float ptgpu_glsl_logf(float x) {
    if (!(x > 0.0f))
        return ptgpu_gm_float(0xFF800000u); /* -inf (a NaN or negative never arrives) */
    const uint32_t b = ptgpu_gm_bits(x);
    int e = (int)((b >> 23) & 255u) - 126;                        /* x = m * 2^e, m in [0.5, 1) */
    float m = ptgpu_gm_float((b & 0x007FFFFFu) | 0x3F000000u);
    if (m < 0.707106781186547524f) {
        e -= 1;
        m = m + m - 1.0f;
    } else {
        m = m - 1.0f;
    }
    const float z = m * m;
    float y = 7.0376836292e-2f;
    y = y * m - 1.1514610310e-1f;
    y = y * m + 1.1676998740e-1f;
    y = y * m - 1.2420140846e-1f;
    y = y * m + 1.4249322787e-1f;
    y = y * m - 1.6668057665e-1f;
    y = y * m + 2.0000714765e-1f;
    y = y * m - 2.4999993993e-1f;
    y = y * m + 3.3333331174e-1f;
    y = y * m * z;
    const float fe = (float)e;
    y = y + -2.12194440e-4f * fe;
    y = y + -0.5f * z;
    float r = m + y;
    r = r + 0.693359375f * fe;
    return r;
}

/* cosine of a float in [0, 8) (the shader's theta = 2 pi u, u in [0, 1]) */
PTGPU_GM float ptgpu_glsl_cosf(float x) {
    /* nearest multiple of pi/2, removed in three exactly representable parts */
    const int n = (int)(x * 0.636619772367581343f + 0.5f);
    const float fn = (float)n;
    float r = x - fn * 1.5703125f;
    r = r - fn * 4.837512969970703125e-4f;
    r = r - fn * 7.54978995489188e-8f;
    const float z = r * r;
    /* sin r and cos r on |r| <= pi/4 */
    float s = -1.9515295891e-4f;
    s = s * z + 8.3321608736e-3f;
    s = s * z - 1.6666654611e-1f;
    s = s * z * r + r;
    float c = 2.443315711809948e-5f;
    c = c * z - 1.388731625493765e-3f;
    c = c * z + 4.166664568298827e-2f;
    c = c * z * z - 0.5f * z + 1.0f;
    switch (n & 3) {
    case 0: return c;
    case 1: return -s;
    case 2: return -c;
    default: return s;
    }
}

#endif /* PTGPU_GLSL_MATH_H */
