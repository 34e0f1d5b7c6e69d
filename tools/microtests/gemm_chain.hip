// micro-benchmark: a chain of dependent [128 x 128] Linear layers on one 16-row tile, weights streamed from an L2-resident
// 4 MB arena in the XDL packing -- the regime of the step kernel's GRU / FFN phases -- with
//   (a) 4 waves per workgroup, 2 output tiles per wave (wmmax_pf of tb_device_xdl.hpp, the step kernel's unit), and
//   (b) 8 waves per workgroup (2 per SIMD), 1 output tile per wave.
// Prints cycles per Linear and bytes / clk / CU.  128 workgroups (B = 32 headline) and 256.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "tb_rollout.hpp"
#include "tb_device_xdl.hpp"

using namespace tb;

// one output tile x 128 k: 4 chunks x NPL planes
struct WUnit1 {
    xh8 w[4][NPL];
    f32x4 b;
};
__device__ __forceinline__ const xh8* wfrag1(const xhalf* wpk, int tile, int lane) {
    return reinterpret_cast<const xh8*>(wpk + ((size_t)(tile * 4) * NPL) * 512 + lane * 8);
}
__device__ __forceinline__ void wload1(WUnit1& u, const xhalf* wpk, int tile, int lane) {
    const xh8* pa = wfrag1(wpk, tile, lane);
    TB_SCHED_FENCE();
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int p = 0; p < NPL; ++p) u.w[c][p] = pa[(c * NPL + p) * 64];
    u.b = splat(0.f);
    TB_SCHED_FENCE();
}
__device__ __forceinline__ void wmma1_pf(f32x4& acc, const WUnit1& u, const xhalf* bp, int plane_stride, WUnit1& un, const xhalf* wpk_next,
                                         int tile, int lane) {
    const xh8* pa = wfrag1(wpk_next, tile, lane);
    TB_SCHED_FENCE();
    xh8 x[4][NPL];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int p = 0; p < NPL; ++p) x[c][p] = ldsb8(bp + p * plane_stride + c * 32);
    f32x4 mid = splat(0.f);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#pragma unroll
        for (int p = 0; p < NPL; ++p) un.w[c][p] = pa[(c * NPL + p) * 64];
        if (NPL == 2) {
            mid = mfma_h(u.w[c][0], x[c][P1], mid);
            mid = mfma_h(u.w[c][P1], x[c][0], mid);
        }
        acc = mfma_h(u.w[c][0], x[c][0], acc);
    }
    __builtin_amdgcn_sched_group_barrier(0x100, 4 * NPL, 0);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        if (NPL == 2) {
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    }
    TB_SCHED_FENCE();
    if (NPL == 2) acc += mid * splat(SPLIT_INV);
    un.b = splat(0.f);
}

constexpr int UNIT_HALFS = 8 * 4 * NPL * 512;  // fp16 per packed [128][128] Linear

template <int NT>
__global__ __launch_bounds__(NT) void k_chain(const xhalf* __restrict__ W, int n_lin_arena, int n_lin, float* __restrict__ out, long long* __restrict__ cyc) {
    __shared__ __attribute__((aligned(16))) xhalf PA[NPL * PLANE], PB[NPL * PLANE];
    const int tid = threadIdx.x, wave = wave_of(tid), lane = tid & 63, kq = lane >> 4, m = lane & 15;
    for (int i = tid; i < NPL * PLANE; i += NT) {
        PA[i] = (xhalf)(0.01f * (i % 37));
        PB[i] = (xhalf)0.f;
    }
    __syncthreads();
    const int first = (blockIdx.x * 7) % n_lin_arena;  // workgroups of an XCD walk the same arena, slightly out of phase
    xhalf* cur = PA;
    xhalf* nxt = PB;
    const long long t0 = clock64();
    if (NT == 256) {
        WUnitX u, u2;
        WNextX n0{W + (size_t)first * UNIT_HALFS, nullptr, 2 * wave, 2 * wave + 1, 4, 0};
        wloadx(u, n0, lane);
        for (int i = 0; i < n_lin; ++i) {
            const int nl = (first + i + 1) % n_lin_arena;
            WNextX nn{W + (size_t)nl * UNIT_HALFS, nullptr, 2 * wave, 2 * wave + 1, 4, 0};
            f32x4 acc[2] = {splat(0.f), splat(0.f)};
            wmmax_pf(acc[0], acc[1], u, cur + m * LDP + kq * 8, PLANE, u2, nn, lane);
            planes_store_c(nxt, 2 * wave, lane, relu4(acc[0]) * splat(0.05f));
            planes_store_c(nxt, 2 * wave + 1, lane, relu4(acc[1]) * splat(0.05f));
            __syncthreads();
            u = u2;
            xhalf* t = cur; cur = nxt; nxt = t;
        }
    } else {
        WUnit1 u, u2;
        wload1(u, W + (size_t)first * UNIT_HALFS, wave, lane);
        for (int i = 0; i < n_lin; ++i) {
            const int nl = (first + i + 1) % n_lin_arena;
            f32x4 acc = splat(0.f);
            wmma1_pf(acc, u, cur + m * LDP + kq * 8, PLANE, u2, W + (size_t)nl * UNIT_HALFS, wave, lane);
            planes_store_c(nxt, wave, lane, relu4(acc) * splat(0.05f));
            __syncthreads();
            u = u2;
            xhalf* t = cur; cur = nxt; nxt = t;
        }
    }
    const long long t1 = clock64();
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
    out[blockIdx.x * NT + tid] = (float)cur[(tid * 3) % (NPL * PLANE)];
}

template <int NT>
static void run(const char* name, int n_wg, const xhalf* W, int n_arena, int n_lin, float* out, long long* cyc) {
    hipLaunchKernelGGL(k_chain<NT>, dim3(n_wg), dim3(NT), 0, 0, W, n_arena, 8, out, cyc);
    (void)hipDeviceSynchronize();
    hipLaunchKernelGGL(k_chain<NT>, dim3(n_wg), dim3(NT), 0, 0, W, n_arena, n_lin, out, cyc);
    if (hipDeviceSynchronize() != hipSuccess) { printf("%s failed\n", name); return; }
    std::vector<long long> h(n_wg);
    (void)hipMemcpy(h.data(), cyc, n_wg * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double per = (double)h[n_wg / 2] / n_lin;
    printf("  %-40s wgs %4d : %7.1f cycles / Linear (min %.1f max %.1f)   %5.1f B/clk/CU\n", name, n_wg, per, (double)h[0] / n_lin,
           (double)h[n_wg - 1] / n_lin, UNIT_HALFS * 2.0 / per);
}

int main() {
    const int n_arena = 64;  // 64 Linears x 64 KB (fp16 pairs) = 4 MB, the step kernel's weight set
    std::vector<unsigned short> hw((size_t)n_arena * UNIT_HALFS);
    for (size_t i = 0; i < hw.size(); ++i) hw[i] = 0x1000 + (i * 2654435761u >> 24);  // small positive halfs
    xhalf* W;
    float* out;
    long long* cyc;
    (void)hipMalloc(&W, hw.size() * 2);
    (void)hipMemcpy(W, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
    (void)hipMalloc(&out, 512 * 512 * 4);
    (void)hipMalloc(&cyc, 512 * 8);
    printf("%s, %d KB per Linear\n", NPL == 2 ? "fp16 pairs" : "bf16", UNIT_HALFS * 2 / 1024);
    for (int n_wg : {128, 256}) {
        run<256>("4 waves x 2 tiles (wmmax_pf)", n_wg, W, n_arena, 256, out, cyc);
        run<512>("8 waves x 1 tile", n_wg, W, n_arena, 256, out, cyc);
    }
    return 0;
}
