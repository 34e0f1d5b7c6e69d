// micro-test: fp32-accurate GEMM on the XDL pipe from split-bf16 operands (x = x0 + x1 + x2, 8 bits each) versus the
// fp32 MFMA chain used by k_step, both against an fp64 reference; plus the issue rate of the bf16 MFMA stream with
// VALU fillers (do XDL MFMAs hide VALU on gfx950?).
//   Y^T[16 out][16 agents] = W[16][128] . X^T[128][16]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void split3(float v, __bf16& h, __bf16& m, __bf16& l) {
    h = (__bf16)v;
    const float r1 = v - (float)h;
    m = (__bf16)r1;
    const float r2 = r1 - (float)m;
    l = (__bf16)r2;
}

// mode 0: fp32 MFMA 16x16x4 (k order of k_step: k = kq*32 + 4j + i);  mode 3/6/9: number of bf16 product terms
__global__ void k_gemm(const float* __restrict__ W, const float* __restrict__ X, float* __restrict__ Y, int mode) {
    const int lane = threadIdx.x, kq = lane >> 4, m = lane & 15;
    f32x4 acc = {0, 0, 0, 0};
    if (mode == 0) {
        for (int j = 0; j < 8; ++j)
            for (int i = 0; i < 4; ++i) {
                const int k = kq * 32 + 4 * j + i;
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(W[m * 128 + k], X[m * 128 + k], acc, 0, 0, 0);
            }
    } else if (mode == 20) {
        // fp16 pair WITHOUT the 2^11 scaling of the low part: x = g0 + g1, g1 = fp16(x - g0) (may be an fp16 denormal); all three
        // products in ONE accumulator
        for (int c = 0; c < 4; ++c) {
            f16x8 a[2], b[2];
            for (int e = 0; e < 8; ++e) {
                const int k = c * 32 + kq * 8 + e;
                const float w = W[m * 128 + k], x = X[m * 128 + k];
                a[0][e] = (_Float16)w; a[1][e] = (_Float16)(w - (float)a[0][e]);
                b[0][e] = (_Float16)x; b[1][e] = (_Float16)(x - (float)b[0][e]);
            }
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0], b[1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[1], b[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0], b[0], acc, 0, 0, 0);
        }
    } else if (mode >= 12) {
        // fp16 pair: x = g0 + 2^-11 g1 (g1 = fp16((x - g0) * 2^11)); products g0g0 | (g0g1 + g1g0) * 2^-11 | g1g1 * 2^-22
        f32x4 mid = {0, 0, 0, 0}, lo = {0, 0, 0, 0};
        for (int c = 0; c < 4; ++c) {
            f16x8 a[2], b[2];
            for (int e = 0; e < 8; ++e) {
                const int k = c * 32 + kq * 8 + e;
                const float w = W[m * 128 + k], x = X[m * 128 + k];
                a[0][e] = (_Float16)w; a[1][e] = (_Float16)((w - (float)a[0][e]) * 2048.f);
                b[0][e] = (_Float16)x; b[1][e] = (_Float16)((x - (float)b[0][e]) * 2048.f);
            }
            if (mode >= 13) lo = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[1], b[1], lo, 0, 0, 0);
            mid = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0], b[1], mid, 0, 0, 0);
            mid = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[1], b[0], mid, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0], b[0], acc, 0, 0, 0);
        }
        const float s = 1.0f / 2048.f;
        for (int r = 0; r < 4; ++r) acc[r] = acc[r] + (mid[r] + lo[r] * s) * s;
    } else {
        f32x4 lo = {0, 0, 0, 0};
        for (int c = 0; c < 4; ++c) {
            bf16x8 a[3], b[3];
            for (int e = 0; e < 8; ++e) {
                const int k = c * 32 + kq * 8 + e;
                __bf16 h, mm, l;
                split3(W[m * 128 + k], h, mm, l);
                a[0][e] = h; a[1][e] = mm; a[2][e] = l;
                split3(X[m * 128 + k], h, mm, l);
                b[0][e] = h; b[1][e] = mm; b[2][e] = l;
            }
            // small terms into their own accumulator first, big term last
            if (mode >= 9) {
                lo = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], b[2], lo, 0, 0, 0);
                lo = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[2], lo, 0, 0, 0);
                lo = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], b[1], lo, 0, 0, 0);
            }
            if (mode >= 6) {
                lo = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[2], lo, 0, 0, 0);
                lo = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], b[0], lo, 0, 0, 0);
                lo = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[1], lo, 0, 0, 0);
            }
            lo = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[1], lo, 0, 0, 0);
            lo = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[0], lo, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[0], acc, 0, 0, 0);
        }
        acc += lo;
    }
    for (int r = 0; r < 4; ++r) Y[(kq * 4 + r) * 16 + m] = acc[r];  // Y[out][agent]
}

// issue-rate probe: 24 bf16 MFMAs (two accumulators) with NV v_fma fillers after each
#define MB_A "v_mfma_f32_16x16x32_bf16 %[a], %[x], %[y], %[a]\n"
#define MB_B "v_mfma_f32_16x16x32_bf16 %[b], %[x], %[y], %[b]\n"
#define F1 "v_fma_f32 %[v0], %[v0], %[c], %[d]\n"
#define F2 F1 "v_fma_f32 %[v1], %[v1], %[c], %[d]\n"
#define F3 F2 "v_fma_f32 %[v2], %[v2], %[c], %[d]\n"
#define F4 F3 "v_fma_f32 %[v3], %[v3], %[c], %[d]\n"
#define F6 F4 "v_fma_f32 %[v0], %[v0], %[c], %[d]\nv_fma_f32 %[v1], %[v1], %[c], %[d]\n"
#define BODY(F) MB_A F MB_B F MB_A F MB_B F MB_A F MB_B F MB_A F MB_B F
template <int NV>
__global__ void k_rate(float* out, long long* cyc, int iters) {
    f32x4 a = {0, 0, 0, 0}, b = a;
    bf16x8 x, y;
    for (int e = 0; e < 8; ++e) { x[e] = (__bf16)(threadIdx.x * 1e-3f + e); y[e] = (__bf16)(1.0f + e); }
    float v0 = 1, v1 = 2, v2 = 3, v3 = 4;
    const float c = 0.999f, d = 1e-3f;
    long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        if (NV == 0) asm volatile(BODY("") : [a] "+v"(a), [b] "+v"(b), [v0] "+v"(v0), [v1] "+v"(v1), [v2] "+v"(v2), [v3] "+v"(v3) : [x] "v"(x), [y] "v"(y), [c] "v"(c), [d] "v"(d));
        if (NV == 1) asm volatile(BODY(F1) : [a] "+v"(a), [b] "+v"(b), [v0] "+v"(v0), [v1] "+v"(v1), [v2] "+v"(v2), [v3] "+v"(v3) : [x] "v"(x), [y] "v"(y), [c] "v"(c), [d] "v"(d));
        if (NV == 2) asm volatile(BODY(F2) : [a] "+v"(a), [b] "+v"(b), [v0] "+v"(v0), [v1] "+v"(v1), [v2] "+v"(v2), [v3] "+v"(v3) : [x] "v"(x), [y] "v"(y), [c] "v"(c), [d] "v"(d));
        if (NV == 3) asm volatile(BODY(F3) : [a] "+v"(a), [b] "+v"(b), [v0] "+v"(v0), [v1] "+v"(v1), [v2] "+v"(v2), [v3] "+v"(v3) : [x] "v"(x), [y] "v"(y), [c] "v"(c), [d] "v"(d));
        if (NV == 4) asm volatile(BODY(F4) : [a] "+v"(a), [b] "+v"(b), [v0] "+v"(v0), [v1] "+v"(v1), [v2] "+v"(v2), [v3] "+v"(v3) : [x] "v"(x), [y] "v"(y), [c] "v"(c), [d] "v"(d));
        if (NV == 6) asm volatile(BODY(F6) : [a] "+v"(a), [b] "+v"(b), [v0] "+v"(v0), [v1] "+v"(v1), [v2] "+v"(v2), [v3] "+v"(v3) : [x] "v"(x), [y] "v"(y), [c] "v"(c), [d] "v"(d));
    }
    long long t1 = clock64();
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = a.x + b.y + v0 + v1 + v2 + v3;
}

int main() {
    srand(1);
    std::vector<float> W(16 * 128), X(16 * 128);
    auto u = [] { return (float)rand() / RAND_MAX * 2.f - 1.f; };
    for (auto& w : W) w = u() * 0.088f;
    for (auto& x : X) x = (u() + u() + u()) * 1.2f;
    std::vector<double> ref(256, 0.0);
    std::vector<float> f32seq(256);
    for (int o = 0; o < 16; ++o)
        for (int a = 0; a < 16; ++a) {
            double s = 0; float fs = 0;
            for (int k = 0; k < 128; ++k) { s += (double)W[o * 128 + k] * (double)X[a * 128 + k]; fs = fmaf(W[o * 128 + k], X[a * 128 + k], fs); }
            ref[o * 16 + a] = s; f32seq[o * 16 + a] = fs;
        }
    float *dW, *dX, *dY;
    (void)hipMalloc(&dW, W.size() * 4); (void)hipMalloc(&dX, X.size() * 4); (void)hipMalloc(&dY, 256 * 4);
    (void)hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice);
    auto report = [&](const char* nm, const float* y) {
        double mx = 0, rms = 0, scale = 0;
        for (int i = 0; i < 256; ++i) { const double e = (double)y[i] - ref[i]; mx = fmax(mx, fabs(e)); rms += e * e; scale += ref[i] * ref[i]; }
        printf("%-44s max|err| = %.3e   rms err / rms value = %.3e\n", nm, mx, sqrt(rms / scale));
    };
    report("host fp32 fmaf chain (k order)", f32seq.data());
    const int modes[7] = {0, 3, 6, 9, 12, 13, 20};
    const char* nm[7] = {"fp32 MFMA 16x16x4 (k_step order)", "bf16 split, 3 products", "bf16 split, 6 products", "bf16 split, 9 products", "fp16 pair, 3 products", "fp16 pair, 4 products", "fp16 pair unscaled, 1 accumulator"};
    for (int i = 0; i < 7; ++i) {
        hipLaunchKernelGGL(k_gemm, dim3(1), dim3(64), 0, 0, dW, dX, dY, modes[i]);
        std::vector<float> y(256);
        (void)hipMemcpy(y.data(), dY, 256 * 4, hipMemcpyDeviceToHost);
        report(nm[i], y.data());
    }
    {
        // tiny weights: |w| < 2^-14 makes every hi part an fp16 denormal
        for (auto& w : W) w = u() * 3e-5f;
        for (int o = 0; o < 16; ++o)
            for (int a = 0; a < 16; ++a) {
                double s2 = 0;
                for (int k = 0; k < 128; ++k) s2 += (double)W[o * 128 + k] * (double)X[a * 128 + k];
                ref[o * 16 + a] = s2;
            }
        (void)hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice);
        for (int i : {0, 2, 4, 5, 6}) {
            hipLaunchKernelGGL(k_gemm, dim3(1), dim3(64), 0, 0, dW, dX, dY, modes[i]);
            std::vector<float> y(256);
            (void)hipMemcpy(y.data(), dY, 256 * 4, hipMemcpyDeviceToHost);
            char nm2[96];
            snprintf(nm2, sizeof nm2, "tiny w: %s", nm[i]);
            report(nm2, y.data());
        }
    }
    {
        for (auto& w : W) w = u() * 0.088f;
        for (auto& x : X) x = u() * u() * 0.05f;  // small, wide-dynamic-range inputs (softmax weights, small activations)
        for (int o = 0; o < 16; ++o)
            for (int a = 0; a < 16; ++a) {
                double s2 = 0;
                for (int k = 0; k < 128; ++k) s2 += (double)W[o * 128 + k] * (double)X[a * 128 + k];
                ref[o * 16 + a] = s2;
            }
        (void)hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice);
        (void)hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice);
        for (int i : {0, 4, 6}) {
            hipLaunchKernelGGL(k_gemm, dim3(1), dim3(64), 0, 0, dW, dX, dY, modes[i]);
            std::vector<float> y(256);
            (void)hipMemcpy(y.data(), dY, 256 * 4, hipMemcpyDeviceToHost);
            char nm2[96];
            snprintf(nm2, sizeof nm2, "small x: %s", nm[i]);
            report(nm2, y.data());
        }
    }
    // issue rate
    const int iters = 4000, blocks = 256;
    float* out; long long* cyc;
    (void)hipMalloc(&out, blocks * 512 * 4); (void)hipMalloc(&cyc, blocks * 8 * 8);
    typedef void (*kt)(float*, long long*, int);
    kt ks[6] = {k_rate<0>, k_rate<1>, k_rate<2>, k_rate<3>, k_rate<4>, k_rate<6>};
    const int nv[6] = {0, 1, 2, 3, 4, 6};
    for (int c = 0; c < 6; ++c) {
        hipLaunchKernelGGL(ks[c], dim3(blocks), dim3(256), 0, 0, out, cyc, 10);
        hipLaunchKernelGGL(ks[c], dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
        (void)hipDeviceSynchronize();
        std::vector<long long> h(blocks * 4);
        (void)hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
        double s = 0; for (auto v : h) s += (double)v; s /= h.size();
        printf("bf16 16x16x32 MFMA + %d v_fma each, one wave/SIMD: %6.2f cycles per MFMA\n", nv[c], s / iters / 8);
    }
    return 0;
}
