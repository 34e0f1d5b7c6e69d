// tb_forward: the policy trunk of ONE simulation step as a stand-alone, un-fused call with the reference's signature --
// `TrafficBots.forward(agent_valid, agent_feature, map_valid, map_feature, tl_valid, tl_feature, goal_valid, goal_feature,
// need_weights)` (src/models/traffic_bots.py:163-247) -- including what the fused step kernel never materialises: the head-mean
// attention weights of the LAST layer of the agent->map, agent->traffic-light and agent<->agent blocks (`need_weights=True`,
// src/models/modules/attention.py:115-146, transformer.py:82-95, agent_interaction.py:61-93), which the reference's
// `require_vis_dict` paths consume (waymo_motion.py:122,167,191).
//
// This is the visualisation / debugging path, not the hot path: plain fp32 FMA kernels over row-major copies of the reference's
// own tensors (no operand splitting, no packing), one launch per operator, ~70 launches per call.  It doubles as an on-device
// cross-check of the fused kernel (tests: policy feature and hidden state of a fused step against this path).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>

namespace tb {
namespace fw {

constexpr int H = 128, NH = 4, DH = 32;

// y[r, :] = (x[r, :] - mean) * rstd * g + b   (two-pass variance, eps 1e-5; one wave per row of 128)
__global__ __launch_bounds__(256) void k_ln(const float* __restrict__ x, const float* __restrict__ g, const float* __restrict__ b,
                                            float* __restrict__ y, int rows) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float a0 = x[(size_t)row * H + lane], a1 = x[(size_t)row * H + 64 + lane];
    float s = a0 + a1;
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s * (1.0f / H);
    const float d0 = a0 - mean, d1 = a1 - mean;
    float v = d0 * d0 + d1 * d1;
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    const float rstd = 1.0f / sqrtf(v * (1.0f / H) + 1e-5f);
    y[(size_t)row * H + lane] = d0 * rstd * g[lane] + b[lane];
    y[(size_t)row * H + 64 + lane] = d1 * rstd * g[64 + lane] + b[64 + lane];
}

// y[r, m] (+)= bias[m] + sum_k x[r, k] * W[m, k]   (W row-major [M, ldw], columns [0, K)); flags: 1 = ReLU, 2 = accumulate into y
__global__ __launch_bounds__(128) void k_linear(const float* __restrict__ x, int ldx, const float* __restrict__ W, int ldw,
                                                const float* __restrict__ bias, float* __restrict__ y, int ldy, int rows, int K, int M,
                                                int flags) {
    extern __shared__ float xs[];  // [8][K]
    const int r0 = blockIdx.x * 8;
    for (int i = threadIdx.x; i < 8 * K; i += 128) {
        const int r = r0 + i / K;
        xs[i] = r < rows ? x[(size_t)r * ldx + i % K] : 0.f;
    }
    __syncthreads();
    for (int m = threadIdx.x; m < M; m += 128) {
        const float* w = W + (size_t)m * ldw;
        float acc[8];
        for (int r = 0; r < 8; ++r) acc[r] = 0.f;
        for (int k = 0; k < K; ++k) {
            const float wk = w[k];
            for (int r = 0; r < 8; ++r) acc[r] = fmaf(xs[r * K + k], wk, acc[r]);
        }
        for (int r = 0; r < 8 && r0 + r < rows; ++r) {
            float v = acc[r] + (bias ? bias[m] : 0.f);
            float* py = y + (size_t)(r0 + r) * ldy + m;
            if (flags & 2) v += *py;
            *py = (flags & 1) ? fmaxf(v, 0.f) : v;
        }
    }
}

// One (instance, source row): softmax(q K^T / sqrt(32)) V over T keys, 4 heads.  k / v: [N, T, ldkv] (k at column 0, v at column H of the
// packed projection).  key_valid [N, T]; eye: the key with the source's own index is masked.  A row with no admissible key gives
// out = 0, weights = 0 and no_tgt = 1 (attention.py:101-107,144-146).  w (optional): head-mean probabilities [N, A, T].
__global__ __launch_bounds__(128) void k_attention(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                                                   int ldkv, const uint8_t* __restrict__ key_valid, int A, int T, int eye,
                                                   float* __restrict__ out, float* __restrict__ w, uint8_t* __restrict__ no_tgt) {
    extern __shared__ float sm[];  // q[128] | logits [4][T] | red[8]
    float* qs = sm;
    float* sl = sm + H;
    float* red = sl + NH * T;
    const int a = blockIdx.x, n = blockIdx.y, tid = threadIdx.x;
    const size_t row = (size_t)n * A + a;
    qs[tid] = q[row * H + tid];
    __syncthreads();
    const float scale = 0.17677669529663687f;  // 1 / sqrt(32)
    int any = 0;
    for (int t = tid; t < T; t += 128) {
        const bool ok = key_valid[(size_t)n * T + t] != 0 && !(eye && t == a);
        const float* kr = k + ((size_t)n * T + t) * ldkv;
        for (int h = 0; h < NH; ++h) {
            float acc = 0.f;
            for (int j = 0; j < DH; ++j) acc = fmaf(qs[h * DH + j], kr[h * DH + j], acc);
            sl[h * T + t] = ok ? acc * scale : -INFINITY;
        }
        any |= ok ? 1 : 0;
    }
    any = __syncthreads_or(any);
    if (!any) {
        out[row * H + tid] = 0.f;
        if (w)
            for (int t = tid; t < T; t += 128) w[row * T + t] = 0.f;
        if (tid == 0) no_tgt[row] = 1;
        return;
    }
    if (tid == 0) no_tgt[row] = 0;
    // per head: max and sum over the keys (wave h of the block handles head h)
    {
        const int h = tid >> 5 & 3, l = tid & 31;  // 32 threads per head
        float mx = -INFINITY;
        for (int t = l; t < T; t += 32) mx = fmaxf(mx, sl[h * T + t]);
        for (int o = 16; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 32));
        float s = 0.f;
        for (int t = l; t < T; t += 32) {
            const float e = expf(sl[h * T + t] - mx);  // (-inf -> 0)
            sl[h * T + t] = e;
            s += e;
        }
        for (int o = 16; o > 0; o >>= 1) s += __shfl_xor(s, o, 32);
        if (l == 0) red[h] = 1.0f / s;
    }
    __syncthreads();
    {
        const int h = tid >> 5;
        const float inv = red[h];
        float acc = 0.f;
        for (int t = 0; t < T; ++t) acc = fmaf(sl[h * T + t] * inv, v[((size_t)n * T + t) * ldkv + tid], acc);
        out[row * H + tid] = acc;
    }
    if (w)
        for (int t = tid; t < T; t += 128)
            w[row * T + t] = 0.25f * (sl[t] * red[0] + sl[T + t] * red[1] + sl[2 * T + t] * red[2] + sl[3 * T + t] * red[3]);
}

// x[r, :] = keep[r] ? x[r, :] + (skip[r] ? 0 : d[r, :]) : 0      (residual adds of a transformer layer; skip / keep may be null)
__global__ void k_residual(float* __restrict__ x, const float* __restrict__ d, const uint8_t* __restrict__ skip, const uint8_t* __restrict__ keep,
                           int rows) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)rows * H) return;
    const int r = (int)(i / H);
    const float add = (skip && skip[r]) ? 0.f : d[i];
    x[i] = (keep && !keep[r]) ? 0.f : x[i] + add;
}

// GRU cell (nn.GRU gate order r, z, n): h' = (1 - z) n + z h, zeroed for invalid rows (agent_temporal.py:147-152)
__global__ void k_gru_cell(const float* __restrict__ gi, const float* __restrict__ gh, const float* __restrict__ h, const uint8_t* __restrict__ valid,
                           float* __restrict__ h_new, int rows) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)rows * H) return;
    const int r = (int)(i / H), c = (int)(i % H);
    const float* a = gi + (size_t)r * 3 * H;
    const float* b = gh + (size_t)r * 3 * H;
    const float rg = 1.0f / (1.0f + expf(-(a[c] + b[c])));
    const float zg = 1.0f / (1.0f + expf(-(a[H + c] + b[H + c])));
    const float ng = tanhf(a[2 * H + c] + rg * b[2 * H + c]);
    h_new[i] = valid[r] ? (1.0f - zg) * ng + zg * h[i] : 0.f;
}

// u[r, :] = relu(mask[r] ? pre[r, :] : 0)      (MLP end activation: mask-fill, then ReLU, mlp.py:80-84)
__global__ void k_mask_relu(const float* __restrict__ pre, const uint8_t* __restrict__ mask, float* __restrict__ u, int rows, int relu) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)rows * H) return;
    const float v = mask[i / H] ? pre[i] : 0.f;
    u[i] = relu ? fmaxf(v, 0.f) : v;
}

// AddLatentGoal tail (add_latent_goal.py:70-77): x = x_valid ? (z_valid ? hh : 0) + x : 0
__global__ void k_fuse_tail(float* __restrict__ x, const float* __restrict__ hh, const uint8_t* __restrict__ z_valid, const uint8_t* __restrict__ x_valid,
                            int rows) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)rows * H) return;
    const int r = (int)(i / H);
    x[i] = x_valid[r] ? (z_valid[r] ? hh[i] : 0.f) + x[i] : 0.f;
}

// interaction bypass (agent_interaction.py:61-77): instances with exactly one valid agent keep the block's input; their weights are 0
__global__ void k_single_agent(const uint8_t* __restrict__ valid, int A, uint8_t* __restrict__ single) {
    const int n = blockIdx.x;
    int c = 0;
    for (int a = threadIdx.x; a < A; a += blockDim.x) c += valid[(size_t)n * A + a] ? 1 : 0;
    c = __syncthreads_count(c);  // (counts threads with c != 0: at most one agent per thread when A <= blockDim.x)
    if (threadIdx.x == 0) single[n] = c == 1;
}
__global__ void k_select_rows(float* __restrict__ y, const float* __restrict__ x, const uint8_t* __restrict__ single, int A, int W, float fill_mode) {
    // y[n, a, :] = single[n] ? (fill_mode < 0 ? x[n, a, :] : fill_mode) : y[n, a, :]
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int n = blockIdx.y;
    if (i >= (size_t)A * W || !single[n]) return;
    const size_t j = (size_t)n * A * W + i;
    y[j] = fill_mode < 0 ? x[j] : fill_mode;
}

}  // namespace fw
}  // namespace tb

// ---------------------------------------------------------------------------------------------------------------------
// host side: the launch sequence (mirrors oracle/trafficbots_oracle.py::policy_step, i.e. traffic_bots.py:205-241)
// ---------------------------------------------------------------------------------------------------------------------
#include <algorithm>
#include <map>
#include <string>
#include "../../include/trafficbots_hip.h"

namespace tb {

size_t forward_scratch_floats(const tb_forward_io* io) {
    const size_t R = (size_t)io->n_inst * io->n_agent;
    const size_t tmax = (size_t)io->n_inst * (size_t)std::max(std::max(io->n_pl, io->n_tl), io->n_agent);
    return R * 128 * 10 + R * 384 * 2 + tmax * 128 + tmax * 256 + R + io->n_inst + 1024;
}

// raw: reference state_dict name -> device pointer of the row-major fp32 tensor.  Returns nullptr on success, else the missing key.
const char* run_forward(const std::map<std::string, const float*>& raw, const tb_forward_io* io, float* scratch, hipStream_t s) {
    using namespace fw;
    const int N = io->n_inst, A = io->n_agent, R = N * A;
    static thread_local std::string missing;
    bool ok = true;
    auto W = [&](const std::string& n) -> const float* {
        auto it = raw.find(n);
        if (it == raw.end()) {
            if (ok) missing = n;
            ok = false;
            return nullptr;
        }
        return it->second;
    };
    float* p = scratch;
    auto take = [&](size_t n) { float* r = p; p += (n + 3) & ~size_t(3); return r; };
    const size_t tmax = (size_t)N * (size_t)std::max(std::max(io->n_pl, io->n_tl), A);
    float* s_ln = take((size_t)R * H); float* q = take((size_t)R * H); float* att = take((size_t)R * H); float* o = take((size_t)R * H);
    float* ffh = take((size_t)R * H); float* x0 = take((size_t)R * H); float* u = take((size_t)R * H); float* hh = take((size_t)R * H);
    float* g1 = take((size_t)R * H); float* g2 = take((size_t)R * H);
    float* gi = take((size_t)R * 3 * H); float* gh = take((size_t)R * 3 * H);
    float* t_ln = take(tmax * H); float* kv = take(tmax * 2 * H);
    uint8_t* no_tgt = reinterpret_cast<uint8_t*>(take((size_t)(R + 3) / 4 + 1));
    uint8_t* single = reinterpret_cast<uint8_t*>(take((size_t)(N + 3) / 4 + 1));
    float* x = io->policy_feature;
    const uint8_t* av = io->agent_valid;
    const dim3 ew((unsigned)(((size_t)R * H + 255) / 256));

    auto ln = [&](const float* src, const std::string& pre, float* dst, int rows) {
        const float* g = W(pre + ".weight"); const float* b = W(pre + ".bias");
        if (ok) hipLaunchKernelGGL(k_ln, dim3((rows + 3) / 4), dim3(256), 0, s, src, g, b, dst, rows);
    };
    auto lin = [&](const float* src, int ldx, const float* w, int ldw, const float* b, float* dst, int ldy, int rows, int K, int M, int flags) {
        if (ok) hipLaunchKernelGGL(k_linear, dim3((rows + 7) / 8), dim3(128), 8 * K * sizeof(float), s, src, ldx, w, ldw, b, dst, ldy, rows, K, M, flags);
    };
    // one pre-LN TransformerCrossAttention layer (transformer.py:189-239); wout: head-mean weights of THIS layer or nullptr
    auto layer = [&](const std::string& pre, const float* tgt, const uint8_t* tgt_valid, int T, int eye, float* wout) {
        const float* w_in = W(pre + ".attn.in_proj_weight"); const float* b_in = W(pre + ".attn.in_proj_bias");
        ln(x, pre + ".norm1", s_ln, R);
        lin(s_ln, H, w_in, H, b_in, q, H, R, H, H, 0);
        ln(tgt, pre + ".norm_tgt", t_ln, N * T);
        lin(t_ln, H, ok ? w_in + (size_t)H * H : nullptr, H, ok ? b_in + H : nullptr, kv, 2 * H, N * T, H, 2 * H, 0);
        if (ok) {
            // up to 8192 keys: 4 x T logits in LDS = 131.6 KB of gfx950's 160 KB -- above the 64 KB a kernel gets without asking
            const size_t lds = (H + NH * T + 8) * sizeof(float);
            // (set on every call that needs it: the attribute belongs to the CURRENT device's copy of the function, a process-wide
            // "done" flag would leave a second GPU of the process without it; the call is a host-side table update)
            if (lds > 48 * 1024)
                ok = hipFuncSetAttribute(reinterpret_cast<const void*>(k_attention), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)((H + NH * 8192 + 8) * sizeof(float))) == hipSuccess;
            if (ok) hipLaunchKernelGGL(k_attention, dim3(A, N), dim3(128), lds, s, q, kv, kv + H, 2 * H, tgt_valid, A, T, eye, att, wout, no_tgt);
        }
        lin(att, H, W(pre + ".attn.out_proj_weight"), H, W(pre + ".attn.out_proj_bias"), o, H, R, H, H, 0);
        if (ok) hipLaunchKernelGGL(k_residual, ew, dim3(256), 0, s, x, o, no_tgt, (const uint8_t*)nullptr, R);
        ln(x, pre + ".norm2", s_ln, R);
        lin(s_ln, H, W(pre + ".linear1.weight"), H, W(pre + ".linear1.bias"), ffh, H, R, H, H, 1);
        lin(ffh, H, W(pre + ".linear2.weight"), H, W(pre + ".linear2.bias"), o, H, R, H, H, 0);
        if (ok) hipLaunchKernelGGL(k_residual, ew, dim3(256), 0, s, x, o, (const uint8_t*)nullptr, av, R);
    };
    (void)hipMemcpyAsync(x, io->agent_feature, (size_t)R * H * sizeof(float), hipMemcpyDeviceToDevice, s);
    for (int l = 0; l < 3; ++l)
        layer("model.transformer_as2pl.layers." + std::to_string(l), io->map_feature, io->map_valid, io->n_pl, 0, l == 2 ? io->attn_pl : nullptr);
    for (int l = 0; l < 3; ++l)
        layer("model.transformer_as2tl.layers." + std::to_string(l), io->tl_feature, io->tl_valid, io->n_tl, 0, l == 2 ? io->attn_tl : nullptr);
    // agent interaction: tgt = the block's INPUT for all three layers, eye mask, single-agent instances bypass (agent_interaction.py:51-93)
    (void)hipMemcpyAsync(x0, x, (size_t)R * H * sizeof(float), hipMemcpyDeviceToDevice, s);
    for (int l = 0; l < 3; ++l)
        layer("model.agent_interaction.transformer.layers." + std::to_string(l), x0, av, A, 1, l == 2 ? io->attn_agent : nullptr);
    hipLaunchKernelGGL(k_single_agent, dim3(N), dim3(256), 0, s, av, A, single);
    hipLaunchKernelGGL(k_select_rows, dim3((A * H + 255) / 256, N), dim3(256), 0, s, x, x0, single, A, H, -1.0f);
    if (io->attn_agent) hipLaunchKernelGGL(k_select_rows, dim3((A * A + 255) / 256, N), dim3(256), 0, s, io->attn_agent, (const float*)nullptr, single, A, A, 0.0f);
    // 3-layer GRU, one step (agent_temporal.py:147-152); hidden [3, N*A, 128] in / out
    const float* inp = x;
    for (int l = 0; l < 3; ++l) {
        const std::string pre = "model.agent_temporal.rnn.";
        float* hl = io->hidden + (size_t)l * R * H;
        lin(inp, H, W(pre + "weight_ih_l" + std::to_string(l)), H, W(pre + "bias_ih_l" + std::to_string(l)), gi, 3 * H, R, H, 3 * H, 0);
        lin(hl, H, W(pre + "weight_hh_l" + std::to_string(l)), H, W(pre + "bias_hh_l" + std::to_string(l)), gh, 3 * H, R, H, 3 * H, 0);
        if (ok) hipLaunchKernelGGL(k_gru_cell, ew, dim3(256), 0, s, gi, gh, hl, av, hl, R);
        inp = hl;
    }
    (void)hipMemcpyAsync(x, inp, (size_t)R * H * sizeof(float), hipMemcpyDeviceToDevice, s);  // (already zero for invalid rows)
    // add_goal (add_latent_goal.py:57-77): mlp_in = 3 x (Linear, LayerNorm, ReLU) with the end activation applied after the mask
    if (io->goal_feature && io->goal_valid) {
        const std::string pi = "model.add_goal.mlp_in.fc_layers.", po = "model.add_goal.mlp_out.fc_layers.";
        lin(io->goal_feature, H, W(pi + "0.weight"), H, W(pi + "0.bias"), g1, H, R, H, H, 0);
        ln(g1, pi + "1", g2, R);
        if (ok) hipLaunchKernelGGL(k_mask_relu, ew, dim3(256), 0, s, g2, io->goal_valid, g1, R, 1);  // (hidden layers: plain ReLU; rows masked at the end anyway)
        lin(g1, H, W(pi + "4.weight"), H, W(pi + "4.bias"), g2, H, R, H, H, 0);
        ln(g2, pi + "5", g1, R);
        if (ok) hipLaunchKernelGGL(k_mask_relu, ew, dim3(256), 0, s, g1, io->goal_valid, g2, R, 1);
        lin(g2, H, W(pi + "8.weight"), H, W(pi + "8.bias"), g1, H, R, H, H, 0);
        ln(g1, pi + "9", g2, R);
        if (ok) hipLaunchKernelGGL(k_mask_relu, ew, dim3(256), 0, s, g2, io->goal_valid, u, R, 1);
        lin(x, H, W(po + "0.weight"), 2 * H, W(po + "0.bias"), hh, H, R, H, H, 0);
        lin(u, H, ok ? W(po + "0.weight") + H : nullptr, 2 * H, nullptr, hh, H, R, H, H, 2 | 1);
        lin(hh, H, W(po + "3.weight"), H, W(po + "3.bias"), g1, H, R, H, H, 1);
        if (ok) hipLaunchKernelGGL(k_fuse_tail, ew, dim3(256), 0, s, x, g1, io->goal_valid, av, R);
    }
    // add_latent: mlp_in = Linear(16 -> 128), ReLU, Linear(128 -> 128), end activation after the mask (z_valid = agent_valid)
    {
        const std::string pi = "model.add_latent.mlp_in.fc_layers.", po = "model.add_latent.mlp_out.fc_layers.";
        lin(io->latent_sample, 16, W(pi + "0.weight"), 16, W(pi + "0.bias"), g1, H, R, 16, H, 1);
        lin(g1, H, W(pi + "3.weight"), H, W(pi + "3.bias"), g2, H, R, H, H, 0);
        if (ok) hipLaunchKernelGGL(k_mask_relu, ew, dim3(256), 0, s, g2, av, u, R, 1);
        lin(x, H, W(po + "0.weight"), 2 * H, W(po + "0.bias"), hh, H, R, H, H, 0);
        lin(u, H, ok ? W(po + "0.weight") + H : nullptr, 2 * H, nullptr, hh, H, R, H, H, 2 | 1);
        lin(hh, H, W(po + "3.weight"), H, W(po + "3.bias"), g1, H, R, H, H, 1);
        if (ok) hipLaunchKernelGGL(k_fuse_tail, ew, dim3(256), 0, s, x, g1, av, av, R);
    }
    return ok ? nullptr : missing.c_str();
}

}  // namespace tb
