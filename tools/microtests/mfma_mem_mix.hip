// micro-benchmark: cost of VMEM / LDS instructions issued between v_mfma_f32_16x16x4_f32 (one wave per SIMD and two).
// Each loop iteration = 16 MFMAs (two chains) with one memory instruction after every 4th MFMA (4 per iteration),
// the unit pattern of wmma_pf.  Buffers are tiny (L1/L2 resident): this measures ISSUE cost, not latency.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MA "v_mfma_f32_16x16x4_f32 %[a], %[x], %[y], %[a]\n"
#define MB "v_mfma_f32_16x16x4_f32 %[b], %[x], %[y], %[b]\n"
#define M4 MA MB MA MB
template <int MODE>
__global__ void k(float* out, long long* cyc, const float* gsrc, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = i;
    __syncthreads();
    f32x4 a = {0, 0, 0, 0}, b = a, r0 = a, r1 = a, r2 = a, r3 = a;
    float x = threadIdx.x * 1e-3f, y = 1.f;
    const float* gp = gsrc + (threadIdx.x & 63) * 4;          // 1 KB per wave, same for all waves
    const float* lp = lds + (threadIdx.x & 63) * 4;
    const unsigned loff = (unsigned)(size_t)lp;               // LDS byte address (low 32 bits of the generic pointer are not valid: use offset)
    const unsigned lofs = (threadIdx.x & 63) * 16;
    long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {
            asm volatile(M4 M4 M4 M4 : [a] "+v"(a), [b] "+v"(b) : [x] "v"(x), [y] "v"(y));
        } else if (MODE == 1) {  // global_load_dwordx4 with 64-bit VGPR address
            asm volatile(M4 "global_load_dwordx4 %[r0], %[p], off\n" M4 "global_load_dwordx4 %[r1], %[p], off offset:1024\n" M4
                         "global_load_dwordx4 %[r2], %[p], off offset:2048\n" M4 "global_load_dwordx4 %[r3], %[p], off offset:3072\n"
                         "s_waitcnt vmcnt(0)\n"
                         : [a] "+v"(a), [b] "+v"(b) , [r0] "=&v"(r0), [r1] "=&v"(r1), [r2] "=&v"(r2), [r3] "=&v"(r3): [x] "v"(x), [y] "v"(y), [p] "v"(gp));
        } else if (MODE == 2) {  // global_load_dwordx4 with SGPR base + 32-bit VGPR offset
            asm volatile(M4 "global_load_dwordx4 %[r0], %[p], %[sb]\n" M4 "global_load_dwordx4 %[r1], %[p], %[sb] offset:1024\n" M4
                         "global_load_dwordx4 %[r2], %[p], %[sb] offset:2048\n" M4 "global_load_dwordx4 %[r3], %[p], %[sb] offset:3072\n"
                         "s_waitcnt vmcnt(0)\n"
                         : [a] "+v"(a), [b] "+v"(b), [r0] "=&v"(r0), [r1] "=&v"(r1), [r2] "=&v"(r2), [r3] "=&v"(r3) : [x] "v"(x), [y] "v"(y), [p] "v"(lofs), [sb] "s"(gsrc));
        } else if (MODE == 3) {  // ds_read_b128
            asm volatile(M4 "ds_read_b128 %[r0], %[p]\n" M4 "ds_read_b128 %[r1], %[p] offset:1024\n" M4 "ds_read_b128 %[r2], %[p] offset:2048\n" M4
                         "ds_read_b128 %[r3], %[p] offset:3072\n"
                         "s_waitcnt lgkmcnt(0)\n"
                         : [a] "+v"(a), [b] "+v"(b), [r0] "=&v"(r0), [r1] "=&v"(r1), [r2] "=&v"(r2), [r3] "=&v"(r3) : [x] "v"(x), [y] "v"(y), [p] "v"(lofs));
        } else if (MODE == 4) {  // 2 global loads per 4 MFMA (8 per iteration)
            asm volatile(M4 "global_load_dwordx4 %[r0], %[p], off\nglobal_load_dwordx4 %[r1], %[p], off offset:1024\n" M4
                         "global_load_dwordx4 %[r2], %[p], off offset:2048\nglobal_load_dwordx4 %[r3], %[p], off offset:3072\n" M4
                         "global_load_dwordx4 %[r0], %[p], off\nglobal_load_dwordx4 %[r1], %[p], off offset:1024\n" M4
                         "global_load_dwordx4 %[r2], %[p], off offset:2048\nglobal_load_dwordx4 %[r3], %[p], off offset:3072\n"
                         "s_waitcnt vmcnt(0)\n"
                         : [a] "+v"(a), [b] "+v"(b), [r0] "=&v"(r0), [r1] "=&v"(r1), [r2] "=&v"(r2), [r3] "=&v"(r3) : [x] "v"(x), [y] "v"(y), [p] "v"(gp));
        }
    }
    long long t1 = clock64();
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = a.x + b.y + r0.x + r1.x + r2.x + r3.x + (float)loff;
}
typedef void (*kern_t)(float*, long long*, const float*, int);
int main() {
    const int iters = 4000, blocks = 256;
    float *out, *gsrc; long long* cyc;
    (void)hipMalloc(&out, blocks * 1024 * sizeof(float));
    (void)hipMalloc(&gsrc, 1 << 20);
    (void)hipMemset(gsrc, 0, 1 << 20);
    (void)hipMalloc(&cyc, blocks * 16 * sizeof(long long));
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    kern_t ks[5] = {k<0>, k<1>, k<2>, k<3>, k<4>};
    const char* nm[5] = {"mfma only", "+4 gload (vaddr64)", "+4 gload (saddr+voff)", "+4 ds_read_b128", "+8 gload (vaddr64)"};
    for (int c = 0; c < 5; ++c)
        for (int threads : {256, 512}) {
            hipLaunchKernelGGL(ks[c], dim3(blocks), dim3(threads), 0, 0, out, cyc, gsrc, 10);
            (void)hipEventRecord(e0, 0);
            hipLaunchKernelGGL(ks[c], dim3(blocks), dim3(threads), 0, 0, out, cyc, gsrc, iters);
            (void)hipEventRecord(e1, 0);
            (void)hipDeviceSynchronize();
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            std::vector<long long> h(blocks * (threads / 64));
            (void)hipMemcpy(h.data(), cyc, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
            double s = 0; for (auto v : h) s += (double)v; s /= h.size();
            const int wps = threads / 256;
            printf("%-24s waves/SIMD=%d  ticks/iter/wave=%[p].1f (16 MFMA = 512)  ns per MFMA per SIMD=%6.2f\n", nm[c], wps, s / iters,
                   ms * 1e6 / ((double)iters * 16 * wps));
        }
    return 0;
}
