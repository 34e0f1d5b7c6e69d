// The two samplers of joint_future_pred on the device (VERDICT r03 missing #2).
//
//   tb_latent_sample : DiagGaussian personalities -- `MyDist.sample` (src/models/modules/distributions.py:18-38: mean where the
//                      per-agent `deterministic` flag is set, mean + std * eps elsewhere; both branches of the reference are the same
//                      numbers because x + 0 is exact) and `DiagGaussian.log_prob` (:40-59).  The rollout prologue draws the same way
//                      (k_rollout_init, tb_rollout_io.latent_sample_out) through latent_draw() below.
//   tb_dest_sample   : DestCategorical over the map polylines (:158-201): log-softmax of the masked logits, arg max (first index of the
//                      maximum, as torch.argmax), an inverse-CDF draw from explicit uniforms, and the log-prob of the chosen index.
//                      After `repeat_interleave_` the reference re-creates Categorical(probs=softmax) (:196-199): probs / sum(probs),
//                      logits = log(clamp(probs, eps, 1 - eps)) -- `from_probs` selects that form.
//
// Plain fp32 VALU work: one 16-lane group per agent (latent) / one wavefront per (instance, agent) row (destination), [N, A, 16] and
// [B, A, P] in, a few bytes out; both launches are microseconds and HBM-latency-bound -- they exist so that nothing between
// tb_encode_scene and tb_rollout is ATen arithmetic, not for throughput.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

#include "../../include/trafficbots_hip.h"
#include "tb_sample.hpp"

namespace tb {

__global__ __launch_bounds__(256) void k_latent_sample(tb_latent_sample_io io, const float* __restrict__ log_std) {
    const int i = blockIdx.x * 16 + (threadIdx.x >> 4), d = threadIdx.x & 15;  // (instance, agent) row i, latent dim d
    const int n_row = io.n_scene * io.k_futures * io.n_agent;
    if (i >= n_row) return;
    const int n = i / io.n_agent, a = i - n * io.n_agent, b = n / io.k_futures;
    const float mu = io.mean[((size_t)b * io.n_agent + a) * 16 + d];
    const float ls = log_std[d];
    const bool det = io.eps == nullptr || (io.deterministic != nullptr && io.deterministic[i] != 0);
    float z;
    if (io.forced != nullptr) z = io.forced[(size_t)i * 16 + d];
    else z = latent_draw(mu, det ? 0.f : io.eps[(size_t)i * 16 + d], ls, det);
    if (io.sample != nullptr) io.sample[(size_t)i * 16 + d] = z;
    if (io.log_prob != nullptr) {
        float lp = latent_logp_term(z, mu, ls);
        // the reference sums the 16 terms with one `sum(-1)`; here a 16-lane butterfly (a different order of the same 16 additions)
#pragma unroll
        for (int o = 8; o >= 1; o >>= 1) lp += __shfl_xor(lp, o, 16);
        if (d == 0) io.log_prob[i] = lp;
    }
}

// one wavefront per (instance, agent) row; lane l owns the contiguous chunk [l * per, (l + 1) * per) of the P logits
__global__ __launch_bounds__(64) void k_dest_sample(tb_dest_sample_io io) {
    const int i = blockIdx.x, lane = threadIdx.x;
    const int n = i / io.n_agent, a = i - n * io.n_agent, b = n / io.k_futures;
    const int P = io.n_pl, per = (P + 63) / 64, j0 = lane * per, j1 = min(P, j0 + per);
    const float* lg = io.dest_logits + ((size_t)b * io.n_agent + a) * P;
    // ---- log-softmax (torch.softmax / Categorical(logits=): subtract the max, exponentiate, sum)
    float m = -INFINITY;
    for (int j = j0; j < j1; ++j) m = fmaxf(m, lg[j]);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    float s = 0.f;
    for (int j = j0; j < j1; ++j) s += expf(lg[j] - m);
    float chunk = s;  // this lane's share of the un-normalised mass (kept for the CDF walk)
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
    const float lse = m + logf(s);
    // ---- arg max: first index of the largest probability (exp is monotone: of the largest logit)
    int best = P;
    for (int j = j0; j < j1; ++j)
        if (lg[j] == m) { best = j; break; }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) best = min(best, __shfl_xor(best, o, 64));
    int pick = best;
    const bool det = io.uniform == nullptr || (io.deterministic != nullptr && io.deterministic[i] != 0);
    if (io.forced != nullptr) {
        pick = io.forced[i];
    } else if (!det) {
        // ---- inverse CDF on the un-normalised masses e_j = exp(l_j - m): the smallest j with  e_0 + ... + e_j > u * sum  (u in [0, 1));
        // exclusive scan of the lanes' chunk sums, then a walk inside the one chunk that straddles the target
        const float target = io.uniform[i] * s;
        float incl = chunk;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const float up = __shfl_up(incl, o, 64);
            if (lane >= o) incl += up;
        }
        const float excl = incl - chunk;
        int found = P;
        if (target >= excl && target < incl) {
            float c = excl;
            for (int j = j0; j < j1; ++j) {
                const float e = expf(lg[j] - m);
                if (e > 0.f) found = j;  // (a walk that ends a hair below its scanned sum keeps the chunk's last polyline with mass)
                c += e;
                if (c > target) break;
            }
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) found = min(found, __shfl_xor(found, o, 64));
        // (u * sum rounded up to the total: no chunk straddles it -- the last polyline that carries mass)
        if (found >= P) {
            int last = -1;
            for (int j = j0; j < j1; ++j)
                if (lg[j] > -INFINITY) last = j;
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) last = max(last, __shfl_xor(last, o, 64));
            found = last;
        }
        pick = found;
    }
    // (a row of NaN logits has no maximum and no mass: best = P, found = -1.  torch.argmax / multinomial would still return an index
    // inside the row, and the sample is used as a polyline index downstream: clamp)
    pick = min(max(pick, 0), P - 1);
    if (lane == 0) {
        if (io.sample != nullptr) io.sample[i] = pick;
        if (io.log_prob != nullptr) {
            float lp = -INFINITY;
            if (pick >= 0 && pick < P) {
                if (io.from_probs) {
                    // Categorical(probs = softmax): probs / probs.sum(-1), then log(clamp(probs, eps, 1 - eps)).  The row sum of the
                    // softmax is 1 up to rounding; its own rounding error is below the tolerance of every consumer, so it is taken as 1.
                    const float pr = expf(lg[pick] - m) / s;
                    lp = logf(fminf(fmaxf(pr, 1.1920929e-07f), 1.f - 1.1920929e-07f));
                } else {
                    lp = lg[pick] - lse;
                }
            }
            io.log_prob[i] = lp;
        }
    }
    if (io.probs != nullptr && n % io.k_futures == 0)  // softmax of the scene's row, written once per scene
        for (int j = j0; j < j1; ++j) io.probs[((size_t)b * io.n_agent + a) * P + j] = expf(lg[j] - m) / s;
}

void launch_latent_sample(const tb_latent_sample_io& io, const float* log_std, hipStream_t s) {
    const int n_row = io.n_scene * io.k_futures * io.n_agent;
    hipLaunchKernelGGL(k_latent_sample, dim3((n_row + 15) / 16), dim3(256), 0, s, io, log_std);
}

void launch_dest_sample(const tb_dest_sample_io& io, hipStream_t s) {
    hipLaunchKernelGGL(k_dest_sample, dim3(io.n_scene * io.k_futures * io.n_agent), dim3(64), 0, s, io);
}

}  // namespace tb
