// micro-test: rows_max / rows_sum (permlane swaps) against __shfl_xor references
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../trafficbots_amd/csrc/tb_device.hpp"
__global__ void k(float* out) {
    const int l = threadIdx.x;
    const float v = (float)l;
    float x, y;
    tb::rows_pair16(v, x, y);
    out[l] = x; out[64 + l] = y;
    tb::rows_pair32(v, x, y);
    out[128 + l] = x; out[192 + l] = y;
    const float w = (float)((l * 37) % 101) - 50.f;
    float m = fmaxf(w, __shfl_xor(w, 16)); m = fmaxf(m, __shfl_xor(m, 32));
    float s = w + __shfl_xor(w, 16); s += __shfl_xor(s, 32);
    out[256 + l] = tb::rows_max(w) - m;
    out[320 + l] = tb::rows_sum(w) - s;
}
int main() {
    float* d; (void)hipMalloc(&d, 384 * 4);
    k<<<1, 64>>>(d);
    float h[384]; (void)hipMemcpy(h, d, 384 * 4, hipMemcpyDeviceToHost);
    const char* names[4] = {"pair16.lo", "pair16.hi", "pair32.lo", "pair32.hi"};
    for (int a = 0; a < 4; ++a) { printf("%s:", names[a]); for (int i = 0; i < 64; i += 4) printf(" %g", h[a * 64 + i]); printf("\n"); }
    float e = 0; for (int i = 256; i < 384; ++i) e = fmaxf(e, fabsf(h[i]));
    printf("rows_max/rows_sum max deviation from shfl reference: %g\n", e);
    return e != 0;
}
