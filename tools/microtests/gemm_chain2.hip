// micro-benchmark (round 5): the step kernel's weight-streaming GEMM chain (gemm_chain.hip, 4 waves x 2 tiles, wmmax_pf) with the
// prefetch distance as the variable:
//   DEPTH 1 = today: unit n+1 is requested at the start of phase n (one unit in flight, the pipe drains once per phase)
//   DEPTH 2 = unit n+2 is requested at the start of phase n (three register buffers)
//   REFILL  = two buffers; the registers of unit n are re-requested for unit n+2 chunk by chunk right behind the MFMAs that consumed them
// Prints cycles per Linear and bytes / clk / CU.  Optional gap of dependent VALU work per phase (argv[1]) standing for epilogues.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "tb_rollout.hpp"
#include "tb_device_xdl.hpp"

using namespace tb;

constexpr int UNIT_HALFS = 8 * 4 * NPL * 512;  // fp16 per packed [128][128] Linear

// REFILL: acc += unit . X^T, and behind the MFMAs of chunk c the same registers are requested again for chunk c of unit `n`
__device__ __forceinline__ void wmmax_refill(f32x4& acc_a, f32x4& acc_b, WUnitX& u, const xhalf* bp, int plane_stride, const WNextX& n, int lane) {
    const xh8* pa = wfragx(n, n.tile_a, lane);
    const xh8* pb = wfragx(n, n.tile_b, lane);
    xh8 x[4][NPL];
    TB_SCHED_FENCE();
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int p = 0; p < NPL; ++p) x[c][p] = ldsb8(bp + p * plane_stride + c * 32);
    f32x4 mid_a = splat(0.f), mid_b = splat(0.f);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        if (NPL == 2) {
            mid_a = mfma_h(u.w[0][c][0], x[c][P1], mid_a);
            mid_b = mfma_h(u.w[1][c][0], x[c][P1], mid_b);
            mid_a = mfma_h(u.w[0][c][P1], x[c][0], mid_a);
            mid_b = mfma_h(u.w[1][c][P1], x[c][0], mid_b);
        }
        acc_a = mfma_h(u.w[0][c][0], x[c][0], acc_a);
        acc_b = mfma_h(u.w[1][c][0], x[c][0], acc_b);
        TB_SCHED_FENCE();
#pragma unroll
        for (int p = 0; p < NPL; ++p) {
            u.w[0][c][p] = pa[(c * NPL + p) * 64];
            u.w[1][c][p] = pb[(c * NPL + p) * 64];
        }
        TB_SCHED_FENCE();
    }
    if (NPL == 2) {
        acc_a += mid_a * splat(SPLIT_INV);
        acc_b += mid_b * splat(SPLIT_INV);
    }
}

template <int MODE>
__global__ __launch_bounds__(256) void k_chain(const xhalf* __restrict__ W, int n_lin_arena, int n_lin, int gap, float* __restrict__ out,
                                               long long* __restrict__ cyc) {
    __shared__ __attribute__((aligned(16))) xhalf PA[NPL * PLANE], PB[NPL * PLANE];
    const int tid = threadIdx.x, wave = wave_of(tid), lane = tid & 63, kq = lane >> 4, m = lane & 15;
    for (int i = tid; i < NPL * PLANE; i += 256) {
        PA[i] = (xhalf)(0.01f * (i % 37));
        PB[i] = (xhalf)0.f;
    }
    __syncthreads();
    const int first = (blockIdx.x * 7) % n_lin_arena;
    xhalf* cur = PA;
    xhalf* nxt = PB;
    auto desc = [&](int i) { return WNextX{W + (size_t)((first + i) % n_lin_arena) * UNIT_HALFS, nullptr, 2 * wave, 2 * wave + 1, 4, 0}; };
    auto finish = [&](f32x4 (&acc)[2]) {
        f32x4 a0 = relu4(acc[0]) * splat(0.05f), a1 = relu4(acc[1]) * splat(0.05f);
        for (int g = 0; g < gap; ++g) {  // dependent VALU chain (epilogue stand-in)
            a0 = a0 * splat(1.0001f) + a1;
            a1 = a1 * splat(0.9999f) + a0 * splat(1e-6f);
        }
        planes_store_c(nxt, 2 * wave, lane, a0);
        planes_store_c(nxt, 2 * wave + 1, lane, a1);
        __syncthreads();
        xhalf* t = cur; cur = nxt; nxt = t;
    };
    const long long t0 = clock64();
    if (MODE == 1) {
        WUnitX u, u2;
        wloadx(u, desc(0), lane);
        for (int i = 0; i < n_lin; i += 2) {
            f32x4 acc[2] = {splat(0.f), splat(0.f)};
            wmmax_pf(acc[0], acc[1], u, cur + m * LDP + kq * 8, PLANE, u2, desc(i + 1), lane);
            finish(acc);
            f32x4 acd[2] = {splat(0.f), splat(0.f)};
            wmmax_pf(acd[0], acd[1], u2, cur + m * LDP + kq * 8, PLANE, u, desc(i + 2), lane);
            finish(acd);
        }
    } else if (MODE == 2) {
        WUnitX a, b, c;
        wloadx(a, desc(0), lane);
        wloadx(b, desc(1), lane);
        for (int i = 0; i < n_lin; i += 3) {
            f32x4 acc[2] = {splat(0.f), splat(0.f)};
            wmmax_pf(acc[0], acc[1], a, cur + m * LDP + kq * 8, PLANE, c, desc(i + 2), lane);
            finish(acc);
            f32x4 acd[2] = {splat(0.f), splat(0.f)};
            wmmax_pf(acd[0], acd[1], b, cur + m * LDP + kq * 8, PLANE, a, desc(i + 3), lane);
            finish(acd);
            f32x4 ace[2] = {splat(0.f), splat(0.f)};
            wmmax_pf(ace[0], ace[1], c, cur + m * LDP + kq * 8, PLANE, b, desc(i + 4), lane);
            finish(ace);
        }
    } else {
        WUnitX a, b;
        wloadx(a, desc(0), lane);
        wloadx(b, desc(1), lane);
        for (int i = 0; i < n_lin; i += 2) {
            f32x4 acc[2] = {splat(0.f), splat(0.f)};
            wmmax_refill(acc[0], acc[1], a, cur + m * LDP + kq * 8, PLANE, desc(i + 2), lane);
            finish(acc);
            f32x4 acd[2] = {splat(0.f), splat(0.f)};
            wmmax_refill(acd[0], acd[1], b, cur + m * LDP + kq * 8, PLANE, desc(i + 3), lane);
            finish(acd);
        }
    }
    const long long t1 = clock64();
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
    out[blockIdx.x * 256 + tid] = (float)cur[(tid * 3) % (NPL * PLANE)];
}

template <int MODE>
static void run(const char* name, int n_wg, const xhalf* W, int n_arena, int n_lin, int gap, float* out, long long* cyc) {
    hipLaunchKernelGGL(k_chain<MODE>, dim3(n_wg), dim3(256), 0, 0, W, n_arena, 12, gap, out, cyc);
    (void)hipDeviceSynchronize();
    hipLaunchKernelGGL(k_chain<MODE>, dim3(n_wg), dim3(256), 0, 0, W, n_arena, n_lin, gap, out, cyc);
    if (hipDeviceSynchronize() != hipSuccess) { printf("%s failed\n", name); return; }
    std::vector<long long> h(n_wg);
    (void)hipMemcpy(h.data(), cyc, n_wg * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double per = (double)h[n_wg / 2] / n_lin;
    std::vector<float> ho(256);
    (void)hipMemcpy(ho.data(), out, 1024, hipMemcpyDeviceToHost);
    double cs = 0;
    for (float v : ho) cs += v;
    printf("  %-44s wgs %4d gap %3d : %7.1f cycles / Linear (min %.1f max %.1f)   %5.1f B/clk/CU   checksum %.6g\n", name, n_wg, gap, per,
           (double)h[0] / n_lin, (double)h[n_wg - 1] / n_lin, UNIT_HALFS * 2.0 / per, cs);
}

int main(int argc, char** argv) {
    const int n_arena = 64;
    std::vector<unsigned short> hw((size_t)n_arena * UNIT_HALFS);
    for (size_t i = 0; i < hw.size(); ++i) hw[i] = 0x1000 + (i * 2654435761u >> 24);
    xhalf* W;
    float* out;
    long long* cyc;
    (void)hipMalloc(&W, hw.size() * 2);
    (void)hipMemcpy(W, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
    (void)hipMalloc(&out, 512 * 512 * 4);
    (void)hipMalloc(&cyc, 512 * 8);
    printf("%s, %d KB per Linear\n", NPL == 2 ? "fp16 pairs" : "bf16", UNIT_HALFS * 2 / 1024);
    for (int gap : {0, 40, 120})
        for (int n_wg : {128, 256}) {
            run<1>("DEPTH 1 (two buffers, today)", n_wg, W, n_arena, 240, gap, out, cyc);
            run<2>("DEPTH 2 (three buffers)", n_wg, W, n_arena, 240, gap, out, cyc);
            run<3>("REFILL (two buffers, refill behind the MFMAs)", n_wg, W, n_arena, 240, gap, out, cyc);
        }
    return 0;
}
