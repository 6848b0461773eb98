// tools/rsq_bias.hip -- how far, and in which direction, is v_rsq_f32 from
// 1/sqrt(x)?  For every float in [1, 4) (two binades = every mantissa at both
// exponent parities): signed error in ulps against the double result.
//   hipcc --offload-arch=gfx950 -O2 tools/rsq_bias.hip -o build/rsq_bias && build/rsq_bias
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>

__global__ void k(double *sum_err, double *sum_abs, unsigned long long *hist, float *worst) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; // 2^24 values
    const uint32_t bits = 0x3F800000u + i;                   // [1, 4)
    const float x = __builtin_bit_cast(float, bits);
    const float r = __builtin_amdgcn_rsqf(x);
    const double exact = 1.0 / sqrt((double)x);
    // ulp of the result (results lie in (0.5, 1])
    const double ulp = (exact > 0.5 ? 0x1.0p-24 : 0x1.0p-25);
    const double e = ((double)r - exact) / ulp;
    atomicAdd(sum_err, e);
    atomicAdd(sum_abs, fabs(e));
    int b = (int)floor(e * 2.0) + 8; // half-ulp bins
    if (b < 0) b = 0;
    if (b > 15) b = 15;
    atomicAdd(&hist[b], 1ull);
    // |x * r^2 - 1| : what a normalised vector's squared length is off by
    (void)worst;
}

int main() {
    double *d_sum, *d_abs;
    unsigned long long *d_hist;
    hipMalloc(&d_sum, 8);
    hipMalloc(&d_abs, 8);
    hipMalloc(&d_hist, 16 * 8);
    hipMemset(d_sum, 0, 8);
    hipMemset(d_abs, 0, 8);
    hipMemset(d_hist, 0, 16 * 8);
    hipLaunchKernelGGL(k, dim3((1u << 24) / 256), dim3(256), 0, 0, d_sum, d_abs, d_hist, nullptr);
    double s, a;
    unsigned long long h[16];
    hipMemcpy(&s, d_sum, 8, hipMemcpyDeviceToHost);
    hipMemcpy(&a, d_abs, 8, hipMemcpyDeviceToHost);
    hipMemcpy(h, d_hist, sizeof h, hipMemcpyDeviceToHost);
    const double n = 16777216.0;
    printf("v_rsq_f32 on [1,4): mean signed error %+.4f ulp, mean |error| %.4f ulp\n", s / n, a / n);
    for (int b = 0; b < 16; ++b)
        if (h[b])
            printf("  error in [%+.1f, %+.1f) ulp: %llu\n", (b - 8) / 2.0, (b - 7) / 2.0, h[b]);
    return 0;
}
