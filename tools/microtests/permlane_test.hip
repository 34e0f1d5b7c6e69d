// micro-test (GPU box): semantics of v_permlane16_swap / v_permlane32_swap and accuracy of the fast exp
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
__device__ __forceinline__ float fast_exp(float x) {
    const float L2E_HI = 1.44269502162933349609375f, L2E_LO = 1.925963033500011e-08f, LN2 = 0.693147182464599609375f;
    const float t = x * L2E_HI;
    float e = fmaf(x, L2E_HI, -t);
    e = fmaf(x, L2E_LO, e);
    const float r = __builtin_amdgcn_exp2f(t);
    return fmaf(r, e * LN2, r);
}
__global__ void k(unsigned* out16, unsigned* out32, float* ex, const float* xs, int n) {
    const unsigned l = threadIdx.x;
    auto r16 = __builtin_amdgcn_permlane16_swap(l, l + 100, false, false);
    auto r32 = __builtin_amdgcn_permlane32_swap(l, l + 100, false, false);
    out16[l] = r16[0]; out16[64 + l] = r16[1];
    out32[l] = r32[0]; out32[64 + l] = r32[1];
    for (int i = l; i < n; i += 64) ex[i] = fast_exp(xs[i]);
}
int main() {
    unsigned *d16, *d32; float *dx, *de; const int n = 1 << 16;
    hipMalloc(&d16, 512); hipMalloc(&d32, 512); hipMalloc(&dx, n * 4); hipMalloc(&de, n * 4);
    float* hx = new float[n]; for (int i = 0; i < n; ++i) hx[i] = -100.0f * (float)i / n;
    hipMemcpy(dx, hx, n * 4, hipMemcpyHostToDevice);
    k<<<1, 64>>>(d16, d32, de, dx, n);
    unsigned h16[128], h32[128]; float* he = new float[n];
    hipMemcpy(h16, d16, 512, hipMemcpyDeviceToHost); hipMemcpy(h32, d32, 512, hipMemcpyDeviceToHost); hipMemcpy(he, de, n * 4, hipMemcpyDeviceToHost);
    printf("permlane16_swap(old=l, src=l+100): ret0 ="); for (int i = 0; i < 64; i += 8) printf(" %u", h16[i]); printf("\n                                   ret1 ="); for (int i = 0; i < 64; i += 8) printf(" %u", h16[64 + i]);
    printf("\npermlane32_swap(old=l, src=l+100): ret0 ="); for (int i = 0; i < 64; i += 8) printf(" %u", h32[i]); printf("\n                                   ret1 ="); for (int i = 0; i < 64; i += 8) printf(" %u", h32[64 + i]);
    double maxrel = 0; for (int i = 0; i < n; ++i) { double r = exp((double)hx[i]); if (r > 1e-37) { double e = fabs(he[i] - r) / r; if (e > maxrel) maxrel = e; } }
    printf("\nfast_exp max rel err on [-100,0] (normal range) = %.3e (fp32 eps = 5.96e-8)\n", maxrel);
    return 0;
}
