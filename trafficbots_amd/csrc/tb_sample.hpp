// Device functions shared by the stand-alone samplers (tb_sample_kernels.hip) and the rollout prologue (k_rollout_init).
#pragma once
#include <hip/hip_runtime.h>

namespace tb {

// `MyDist.sample` for a diagonal Gaussian (src/models/modules/distributions.py:18-38): the mean where the agent is deterministic,
// `Normal.rsample` = loc + eps * scale elsewhere (a multiply and an add, separately rounded: the build has -ffp-contract=off).
__device__ __forceinline__ float latent_draw(float mu, float eps, float log_std, bool deterministic) {
    return deterministic ? mu : mu + eps * expf(log_std);
}

// one dimension of `Independent(Normal).log_prob` (:40-59): -(z - mu)^2 / (2 var) - log(std) - log(sqrt(2 pi))
__device__ __forceinline__ float latent_logp_term(float z, float mu, float log_std) {
    const float stdv = expf(log_std);
    const float diff = z - mu;
    return -(diff * diff) / (2.f * (stdv * stdv)) - logf(stdv) - 0.9189385332046727f;
}

}  // namespace tb
