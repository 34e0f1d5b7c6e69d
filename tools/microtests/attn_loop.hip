// micro-benchmark of the step kernel's attention loop (tb_device_xdl.hpp: attention_head_x and candidate variants) under the
// real kernel's conditions: 4 waves per workgroup (wave = head), one workgroup per 16-row tile, the row tiles of a group share
// one K/V set and start their key walk at staggered blocks, K/V fragment-major in global memory (L2 / HBM resident, one set per
// group and "layer"), 128 or 256 workgroups.  Prints cycles per 32-key block for each variant; the results of the variants are
// cross-checked against variant 0 (same arithmetic up to the order of the online-softmax rescaling).
//
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form [-DTB_XDL_BF16] \
//        -I trafficbots_amd/csrc tools/microtests/attn_loop.hip -o tools/microtests/bin/attn_loop[_bf16]
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <vector>

#include "tb_rollout.hpp"
#include "tb_device_xdl.hpp"
#include "tb_step_common.hpp"

using namespace tb;

#include "attn_variants.hpp"

template <int VAR, int NT = NTHREADS>
__global__ __launch_bounds__(NT) void k_attn(const xhalf* __restrict__ Kall, const xhalf* __restrict__ Vall, const float* __restrict__ kbias,
                                                   const float* __restrict__ wdummy, int n_key_pad, int n_layers, int reps, float* __restrict__ out,
                                                   long long* __restrict__ cyc, int kv_share) {
    const int tid = threadIdx.x, wave8 = wave_of(tid), wave = wave8 & 3, lane = tid & 63, kq = lane >> 4, m = lane & 15;
    if (NT == 512) n_key_pad >>= 1;  // each wave of a pair walks half of the keys
    int g, rt;
    step_tile_map(g, rt);  // the row tiles of a group on ONE XCD, as in k_step_x
    const int n_rt = gridDim.x;
    const size_t ls = (size_t)(NT == 512 ? 2 * n_key_pad : n_key_pad) * H;  // fp16 per plane per (group, layer); K and V each hold NPL planes
    const int kstart = ((rt * (n_key_pad >> 5)) / n_rt) << 5;
    f32x4 q[2];
    q[0] = f32x4{0.01f * (m + 1), -0.02f * (kq + 1), 0.03f * wave, 0.005f * (lane & 7)};
    q[1] = f32x4{-0.01f * (m + 2), 0.015f * (kq + 2), 0.02f, -0.004f * (lane & 3)};
    f32x4 acc[2] = {splat(0.f), splat(0.f)};
    WUnitX un;
    const WNextX nx = wstdx(wdummy, 0, nullptr, wave);
    long long t_total = 0;
    for (int r = 0; r < reps; ++r) {
        for (int l = 0; l < n_layers; ++l) {
            const int gk = kv_share ? 0 : g;
            const size_t half_off = (NT == 512 && wave8 >= 4) ? (size_t)(n_key_pad >> 5) * KV_BLOCK_HALFS : 0;
            const xhalf* K0 = Kall + ((size_t)gk * n_layers + l) * NPL * ls + half_off;
            const xhalf* V0 = Vall + ((size_t)gk * n_layers + l) * NPL * ls + half_off;
            const float* kb = kbias + (size_t)g * n_key_pad;
            AttnPreX pre;
            f32x4 o[2];
            __syncthreads();
            const long long t0 = clock64();
            if (VAR == 0) attention_prefetch_x(pre, K0, V0, kb, n_key_pad, kstart, wave, lane);
            bool nov;
            if (VAR == 0) nov = attention_head_x<false>(q, pre, K0, V0, kb, n_key_pad, kstart, wave, lane, -1, o, un, nx);
            else if (VAR == 4) nov = attention_head_dual(q, K0, V0, kb, n_key_pad, kstart, wave, lane, o, un, nx);
            else nov = attention_head_var<VAR == 4 ? 1 : VAR>(q, pre, K0, V0, kb, n_key_pad, kstart, wave, lane, o, un, nx);
            const long long t1 = clock64();
            t_total += t1 - t0;
            acc[0] += o[0] + splat(nov ? 1.f : 0.f) + un.w[0][0][0][0] * 0.f;
            acc[1] += o[1];
            q[0] += o[0] * splat(1e-3f);  // dependent chain across layers, as in the real kernel
        }
    }
    if (lane == 0 && wave8 < 4) cyc[(blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave] = t_total;
    if (wave8 >= 4) return;
    float* po = out + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * NTHREADS + tid) * 8;
    for (int i = 0; i < 4; ++i) {
        po[i] = acc[0][i];
        po[4 + i] = acc[1][i];
    }
}

template <int VAR, int NT = NTHREADS>
static double run(const char* name, int n_rt, int n_group, int n_key_pad, int n_layers, int reps, const xhalf* K, const xhalf* V, const float* kb,
                  const float* wd, float* out, long long* cyc, std::vector<float>* res, int kv_share = 0) {
    dim3 grid(n_rt, n_group);
    hipLaunchKernelGGL((k_attn<VAR, NT>), grid, dim3(NT), 0, 0, K, V, kb, wd, n_key_pad, n_layers, 1, out, cyc, kv_share);  // warm
    (void)hipDeviceSynchronize();
    hipLaunchKernelGGL((k_attn<VAR, NT>), grid, dim3(NT), 0, 0, K, V, kb, wd, n_key_pad, n_layers, reps, out, cyc, kv_share);
    if (hipDeviceSynchronize() != hipSuccess) {
        printf("%s: launch failed\n", name);
        return 0;
    }
    const int nw = n_rt * n_group * 4;
    std::vector<long long> h(nw);
    (void)hipMemcpy(h.data(), cyc, nw * sizeof(long long), hipMemcpyDeviceToHost);
    std::vector<long long> s = h;
    std::sort(s.begin(), s.end());
    const double blocks = (double)reps * n_layers * (n_key_pad / 32);
    std::vector<float> o((size_t)n_rt * n_group * NTHREADS * 8);
    (void)hipMemcpy(o.data(), out, o.size() * sizeof(float), hipMemcpyDeviceToHost);
    double maxdiff = 0;
    bool finite = true;
    for (size_t i = 0; i < o.size(); ++i) {
        finite &= std::isfinite(o[i]);
        if (res && !res->empty()) maxdiff = std::fmax(maxdiff, std::fabs(o[i] - (*res)[i]));
    }
    if (res && res->empty()) *res = o;
    printf("  %-44s wgs %4d keys %5d : %7.1f cycles / 32-key block (median wave; min %.1f max %.1f)  finite %d  max|d vs v0| %.2e\n", name,
           n_rt * n_group, n_key_pad, s[nw / 2] / blocks, s[0] / blocks, s[nw - 1] / blocks, (int)finite, maxdiff);
    return s[nw / 2] / blocks;
}

int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 20;
    struct Cfg {
        int n_rt, n_group, keys, layers;
    } cfgs[] = {{4, 32, 256, 3}, {8, 32, 1024, 3}, {4, 32, 128, 3}};
    for (const Cfg& c : cfgs) {
        const size_t ls = (size_t)c.keys * H;
        const size_t nk = (size_t)c.n_group * c.layers * NPL * ls;
        std::vector<unsigned short> hk(nk), hv(nk);
        srand(1);
        for (size_t i = 0; i < nk; ++i) {
#ifdef TB_XDL_BF16
            float a = (rand() % 2001 - 1000) * 1e-3f, b = (rand() % 2001 - 1000) * 1e-3f;
            unsigned ua, ub;
            memcpy(&ua, &a, 4);
            memcpy(&ub, &b, 4);
            hk[i] = ua >> 16;
            hv[i] = ub >> 16;
#else
            _Float16 a = (_Float16)((rand() % 2001 - 1000) * 1e-3f), b = (_Float16)((rand() % 2001 - 1000) * 1e-3f);
            memcpy(&hk[i], &a, 2);
            memcpy(&hv[i], &b, 2);
#endif
        }
        xhalf *K, *V;
        float *kb, *wd, *out;
        long long* cyc;
        (void)hipMalloc(&K, nk * 2);
        (void)hipMalloc(&V, nk * 2);
        (void)hipMemcpy(K, hk.data(), nk * 2, hipMemcpyHostToDevice);
        (void)hipMemcpy(V, hv.data(), nk * 2, hipMemcpyHostToDevice);
        (void)hipMalloc(&kb, (size_t)c.n_group * c.keys * 4);
        (void)hipMemset(kb, 0, (size_t)c.n_group * c.keys * 4);
        (void)hipMalloc(&wd, 1 << 20);
        (void)hipMemset(wd, 0, 1 << 20);
        const size_t nwg = (size_t)c.n_rt * c.n_group;
        (void)hipMalloc(&out, nwg * NTHREADS * 8 * 4);
        (void)hipMalloc(&cyc, nwg * 4 * 8);
        printf("config: %d row tiles x %d groups, %d keys, %d layers, %s\n", c.n_rt, c.n_group, c.keys, c.layers, NPL == 2 ? "fp16 pairs" : "bf16");
        std::vector<float> ref;
        run<0>("v0 attention_head_x (tb_device_xdl.hpp)", c.n_rt, c.n_group, c.keys, c.layers, reps, K, V, kb, wd, out, cyc, &ref);
        run<1>("v1 64 keys per iteration", c.n_rt, c.n_group, c.keys, c.layers, reps, K, V, kb, wd, out, cyc, &ref);
        run<3>("v3 v1 + rescale only when the max moved", c.n_rt, c.n_group, c.keys, c.layers, reps, K, V, kb, wd, out, cyc, &ref);
        run<4>("v4 two softmax chains per wave", c.n_rt, c.n_group, c.keys, c.layers, reps, K, V, kb, wd, out, cyc, &ref);
        run<0, 512>("v0, 8 waves: 2 per head, half the keys each", c.n_rt, c.n_group, c.keys, c.layers, reps, K, V, kb, wd, out, cyc, nullptr);
        run<1, 512>("v1, 8 waves: 2 per head, half the keys each", c.n_rt, c.n_group, c.keys, c.layers, reps, K, V, kb, wd, out, cyc, nullptr);
        run<0>("v0, every group reads group 0's K/V (L2)", c.n_rt, c.n_group, c.keys, c.layers, reps, K, V, kb, wd, out, cyc, nullptr, 1);
        run<1>("v1, every group reads group 0's K/V (L2)", c.n_rt, c.n_group, c.keys, c.layers, reps, K, V, kb, wd, out, cyc, nullptr, 1);
        (void)hipFree(K); (void)hipFree(V); (void)hipFree(kb); (void)hipFree(wd); (void)hipFree(out); (void)hipFree(cyc);
    }
    return 0;
}
