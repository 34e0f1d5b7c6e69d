// micro-benchmark: how much independent VALU work fits in the shadow of v_mfma_f32_16x16x4_f32 on gfx950,
// with one and with two waves per SIMD.  Loop body = 16 MFMAs (two accumulator chains), each followed by K
// independent VALU instructions of one kind; prints s_memtime ticks per MFMA.
//   hipcc --offload-arch=gfx950 -O3 tools/microtests/mfma_valu_overlap.hip -o tools/microtests/bin/mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define MF_A "v_mfma_f32_16x16x4_f32 %0, %10, %11, %0\n"
#define MF_B "v_mfma_f32_16x16x4_f32 %1, %10, %11, %1\n"

#define V0(op)
#define V1(op) op(2)
#define V2(op) op(2) op(3)
#define V3(op) op(2) op(3) op(4)
#define V4(op) op(2) op(3) op(4) op(5)
#define V5(op) op(2) op(3) op(4) op(5) op(6)
#define V6(op) op(2) op(3) op(4) op(5) op(6) op(7)
#define V7(op) op(2) op(3) op(4) op(5) op(6) op(7) op(8)
#define V8(op) op(2) op(3) op(4) op(5) op(6) op(7) op(8) op(9)
#define V10(op) V8(op) op(2) op(3)
#define V12(op) V8(op) op(2) op(3) op(4) op(5)

#define OP_FMA(r) "v_fma_f32 %" #r ", %" #r ", %12, %13\n"
#define OP_EXP(r) "v_exp_f32 %" #r ", %" #r "\n"
#define OP_NOP(r) "s_nop 0\n"

#define BODY(VK, op) \
    MF_A VK(op) MF_B VK(op) MF_A VK(op) MF_B VK(op) MF_A VK(op) MF_B VK(op) MF_A VK(op) MF_B VK(op) \
    MF_A VK(op) MF_B VK(op) MF_A VK(op) MF_B VK(op) MF_A VK(op) MF_B VK(op) MF_A VK(op) MF_B VK(op)

#define KERNEL(name, VK, op)                                                                                   \
    __global__ void name(float* out, long long* cyc, int iters) {                                              \
        f32x4 a = {0, 0, 0, 0}, b = {0, 0, 0, 0};                                                              \
        float x = threadIdx.x * 1e-3f, y = 1.0f + threadIdx.x * 1e-6f;                                         \
        float v0 = x, v1 = x + 1, v2 = x + 2, v3 = x + 3, v4 = x + 4, v5 = x + 5, v6 = x + 6, v7 = x + 7;      \
        const float c = 0.999f, d = 1e-3f;                                                                     \
        long long t0 = clock64();                                                                              \
        for (int i = 0; i < iters; ++i) {                                                                      \
            asm volatile(BODY(VK, op)                                                                          \
                         : "+v"(a), "+v"(b), "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) \
                         : "v"(x), "v"(y), "v"(c), "v"(d));                                                    \
        }                                                                                                      \
        long long t1 = clock64();                                                                              \
        if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;      \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a.x + b.y + v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;       \
    }

KERNEL(k_fma0, V0, OP_FMA)
KERNEL(k_fma1, V1, OP_FMA)
KERNEL(k_fma2, V2, OP_FMA)
KERNEL(k_fma3, V3, OP_FMA)
KERNEL(k_fma4, V4, OP_FMA)
KERNEL(k_fma5, V5, OP_FMA)
KERNEL(k_fma6, V6, OP_FMA)
KERNEL(k_fma7, V7, OP_FMA)
KERNEL(k_fma8, V8, OP_FMA)
KERNEL(k_fma10, V10, OP_FMA)
KERNEL(k_fma12, V12, OP_FMA)
KERNEL(k_exp1, V1, OP_EXP)
KERNEL(k_exp2, V2, OP_EXP)
KERNEL(k_exp4, V4, OP_EXP)
KERNEL(k_nop4, V4, OP_NOP)
KERNEL(k_nop8, V8, OP_NOP)

// VALU only (no MFMA): cycles per VALU instruction
__global__ void k_valu_only(float* out, long long* cyc, int iters) {
    float x = threadIdx.x * 1e-3f;
    float v0 = x, v1 = x + 1, v2 = x + 2, v3 = x + 3, v4 = x + 4, v5 = x + 5, v6 = x + 6, v7 = x + 7;
    const float c = 0.999f, d = 1e-3f;
    f32x4 a = {0, 0, 0, 0}, b = a;
    float y = 1.f;
    long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        asm volatile(V8(OP_FMA) V8(OP_FMA) V8(OP_FMA) V8(OP_FMA) V8(OP_FMA) V8(OP_FMA) V8(OP_FMA) V8(OP_FMA)
                     V8(OP_FMA) V8(OP_FMA) V8(OP_FMA) V8(OP_FMA) V8(OP_FMA) V8(OP_FMA) V8(OP_FMA) V8(OP_FMA)
                     : "+v"(a), "+v"(b), "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7)
                     : "v"(x), "v"(y), "v"(c), "v"(d));
    }
    long long t1 = clock64();
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = a.x + b.y + v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
}

typedef void (*kern_t)(float*, long long*, int);

static void run(const char* name, kern_t k, int per_iter_mfma, int per_iter_valu) {
    const int iters = 2000, blocks = 256;
    float* out;
    long long* cyc;
    (void)hipMalloc(&out, blocks * 1024 * sizeof(float));
    (void)hipMalloc(&cyc, blocks * 16 * sizeof(long long));
    for (int threads : {256, 512, 1024}) {
        hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, out, cyc, 10);
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
        (void)hipEventRecord(e1, 0);
        (void)hipDeviceSynchronize();
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        std::vector<long long> h(blocks * (threads / 64));
        (void)hipMemcpy(h.data(), cyc, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
        double s = 0;
        for (auto v : h) s += (double)v;
        s /= h.size();
        const double per_iter = s / iters;
        const int wps = threads / 256;
        if (per_iter_mfma)
            printf("%-10s waves/SIMD=%d  ticks/iter/wave=%8.1f  ns per MFMA per SIMD=%6.2f  (valu/mfma=%d)\n", name, wps, per_iter,
                   ms * 1e6 / ((double)iters * per_iter_mfma * wps), per_iter_valu / per_iter_mfma);
        else
            printf("%-10s waves/SIMD=%d  ticks/iter/wave=%8.1f  ns per VALU per SIMD=%6.3f\n", name, wps, per_iter,
                   ms * 1e6 / ((double)iters * per_iter_valu * wps));
    }
    (void)hipFree(out);
    (void)hipFree(cyc);
}

int main() {
    run("valu_only", k_valu_only, 0, 128);
    run("fma x0", k_fma0, 16, 0);
    run("fma x1", k_fma1, 16, 16);
    run("fma x2", k_fma2, 16, 32);
    run("fma x3", k_fma3, 16, 48);
    run("fma x4", k_fma4, 16, 64);
    run("fma x5", k_fma5, 16, 80);
    run("fma x6", k_fma6, 16, 96);
    run("fma x7", k_fma7, 16, 112);
    run("fma x8", k_fma8, 16, 128);
    run("fma x10", k_fma10, 16, 160);
    run("fma x12", k_fma12, 16, 192);
    run("exp x1", k_exp1, 16, 16);
    run("exp x2", k_exp2, 16, 32);
    run("exp x4", k_exp4, 16, 64);
    run("nop x4", k_nop4, 16, 64);
    run("nop x8", k_nop8, 16, 128);
    return 0;
}
