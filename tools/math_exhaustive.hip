// Exhaustive check of hand-rolled correctly-rounded sqrt / reciprocal
// candidates against hipcc's IEEE sqrtf and 1.0f/x over all 2^32 inputs.
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/math_exhaustive.hip -o /tmp/mathx
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>

__device__ __forceinline__ float as_f(uint32_t u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ uint32_t as_u(float f) { return __builtin_bit_cast(uint32_t, f); }

__device__ __forceinline__ float sqrt_fast(float x) {
    float s = __builtin_amdgcn_sqrtf(x);
    float dn = as_f(as_u(s) - 1u), up = as_f(as_u(s) + 1u);
    float r_dn = __builtin_fmaf(-dn, s, x);
    float r_up = __builtin_fmaf(-up, s, x);
    s = (r_dn <= 0.0f) ? dn : s;
    s = (r_up > 0.0f) ? up : s;
    return s;
}

template <int STEPS> __device__ __forceinline__ float rcp_fast(float b) {
    float y = __builtin_amdgcn_rcpf(b);
#pragma unroll
    for (int i = 0; i < STEPS; ++i) {
        float e = __builtin_fmaf(-b, y, 1.0f);
        y = __builtin_fmaf(e, y, y);
    }
    return y;
}

__device__ bool same(float a, float b) {
    if (a != a && b != b) return true;  // both NaN
    return as_u(a) == as_u(b);
}

// counters: [0] sqrt mism in fast range, [1] sqrt mism anywhere, [2..4] rcp steps 1..3 mism in range, [5] first bad sqrt, [6..8] first bad rcp
__global__ void check(unsigned long long *out) {
    uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long c[5] = {0, 0, 0, 0, 0};
    for (uint64_t u = tid; u < (1ull << 32); u += stride) {
        float x = as_f((uint32_t)u);
        float ref = sqrtf(x);
        float got = sqrt_fast(x);
        bool ok = same(ref, got);
        if (!ok) { c[1]++; if (x >= 0x1p-96f) { c[0]++; atomicMin(&out[5], (unsigned long long)u); } }
        float ax = fabsf(x);
        bool inrange = ax >= 0x1p-62f && ax <= 0x1p62f;
        float rr = 1.0f / x;
        if (inrange) {
            if (!same(rr, rcp_fast<1>(x))) { c[2]++; atomicMin(&out[6], (unsigned long long)u); }
            if (!same(rr, rcp_fast<2>(x))) { c[3]++; atomicMin(&out[7], (unsigned long long)u); }
            if (!same(rr, rcp_fast<3>(x))) { c[4]++; atomicMin(&out[8], (unsigned long long)u); }
        }
    }
    for (int i = 0; i < 5; ++i) if (c[i]) atomicAdd(&out[i], c[i]);
}

int main() {
    unsigned long long *d, h[9];
    hipMalloc(&d, sizeof h);
    for (int i = 0; i < 9; ++i) h[i] = i >= 5 ? ~0ull : 0ull;
    hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(check, dim3(4096), dim3(256), 0, 0, d);
    hipDeviceSynchronize();
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    printf("sqrt_fast mismatches: x>=2^-96: %llu, anywhere: %llu (first bad in range %#llx)\n", h[0], h[1], h[5]);
    printf("rcp_fast mismatches in |x| in [2^-62,2^62]: 1 step %llu (first %#llx), 2 steps %llu (first %#llx), 3 steps %llu (first %#llx)\n",
           h[2], h[6], h[3], h[7], h[4], h[8]);
    return 0;
}
