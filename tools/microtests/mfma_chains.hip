// micro-benchmark: v_mfma_f32_16x16x4_f32 rate on gfx950 as a function of the number of independent accumulator
// chains per wave and of waves per SIMD; reports s_memtime ticks AND wall-clock (hipEvent) per MFMA per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define M(i) "v_mfma_f32_16x16x4_f32 %" #i ", %8, %9, %" #i "\n"
#define C1 M(0) M(0) M(0) M(0) M(0) M(0) M(0) M(0) M(0) M(0) M(0) M(0) M(0) M(0) M(0) M(0)
#define C2 M(0) M(1) M(0) M(1) M(0) M(1) M(0) M(1) M(0) M(1) M(0) M(1) M(0) M(1) M(0) M(1)
#define C4 M(0) M(1) M(2) M(3) M(0) M(1) M(2) M(3) M(0) M(1) M(2) M(3) M(0) M(1) M(2) M(3)
#define C8 M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7)
#define KERNEL(name, body)                                                                                              \
    __global__ void name(float* out, long long* cyc, int iters) {                                                       \
        f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0, a4 = a0, a5 = a0, a6 = a0, a7 = a0;                           \
        float x = threadIdx.x * 1e-3f, y = 1.0f + threadIdx.x * 1e-6f;                                                  \
        long long t0 = clock64();                                                                                       \
        for (int i = 0; i < iters; ++i)                                                                                 \
            asm volatile(body : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y)); \
        long long t1 = clock64();                                                                                       \
        if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;               \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0.x + a1.x + a2.x + a3.x + a4.x + a5.x + a6.x + a7.x;             \
    }
KERNEL(k1, C1)
KERNEL(k2, C2)
KERNEL(k4, C4)
KERNEL(k8, C8)
typedef void (*kern_t)(float*, long long*, int);
int main() {
    const int iters = 4000, blocks = 256;
    float* out; long long* cyc;
    (void)hipMalloc(&out, blocks * 1024 * sizeof(float));
    (void)hipMalloc(&cyc, blocks * 16 * sizeof(long long));
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    kern_t ks[4] = {k1, k2, k4, k8};
    const int nch[4] = {1, 2, 4, 8};
    for (int c = 0; c < 4; ++c)
        for (int threads : {256, 512, 1024}) {
            hipLaunchKernelGGL(ks[c], dim3(blocks), dim3(threads), 0, 0, out, cyc, 10);
            (void)hipEventRecord(e0, 0);
            hipLaunchKernelGGL(ks[c], dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
            (void)hipEventRecord(e1, 0);
            (void)hipDeviceSynchronize();
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            std::vector<long long> h(blocks * (threads / 64));
            (void)hipMemcpy(h.data(), cyc, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
            double s = 0; for (auto v : h) s += (double)v; s /= h.size();
            const int wps = threads / 256;
            const double n_mfma_simd = (double)iters * 16 * wps;
            const double tf = (double)blocks * 4 * n_mfma_simd * 2048.0 / (ms * 1e-3) / 1e12;
            printf("chains=%d waves/SIMD=%d  ticks/MFMA/SIMD=%6.2f  ns/MFMA/SIMD=%6.2f  ticks/ns=%5.3f  chip rate=%6.1f TFLOP/s\n", nch[c], wps,
                   s / n_mfma_simd, ms * 1e6 / n_mfma_simd, s / (ms * 1e6), tf);
        }
    return 0;
}
