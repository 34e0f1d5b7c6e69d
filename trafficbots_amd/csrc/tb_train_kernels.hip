// Forward training losses of a replayed episode: DifferentiableReward.get over the recorded buffer
// (`src/utils/rewards.py:33-131`) and the six "sum" states of TrainingMetrics.update (`src/models/metrics/training.py:62-139`)
// with BalancedKL's forward value (`src/models/metrics/loss.py:35-74`).  HBM-bound elementwise + reduction work:
//   k_reward         one thread per (scene, agent, step): imitation errors (+ the five-circle collision penalty over the other
//                    agents of the scene when w_collision > 0), writes diffbar_rewards / diffbar_rewards_valid;
//   k_train_partials one thread per (scene, agent): walks the S steps for the reward sums and pred_valid.any, diagonal-Gaussian
//                    KL of the 16 latent dims, negative log-likelihood of the ground-truth destination; block reduction, one
//                    double atomicAdd per field and workgroup.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/trafficbots_hip.h"

namespace tb {

constexpr int TP_FIELDS = 6;

__device__ __forceinline__ float crit_eval(int crit, float a, float b) {
    const float d = __fadd_rn(a, -b);
    if (crit == 1) return __fmul_rn(d, d);  // MSELoss
    const float ad = fabsf(d);
    if (crit == 2) return ad;                                                        // L1Loss
    return ad < 1.f ? __fmul_rn(__fmul_rn(0.5f, ad), ad) : __fadd_rn(ad, -0.5f);     // SmoothL1Loss, beta = 1
}

__device__ __forceinline__ float cast_rad(float x) {  // transform_utils.py:9-11, python's sign-of-divisor remainder
    const float PI_F = 3.14159265358979323846f, TWO_PI_F = 6.28318530717958647692f;
    float r = fmodf(__fadd_rn(x, PI_F), TWO_PI_F);
    if (r != 0.f && r < 0.f) r = __fadd_rn(r, TWO_PI_F);
    return __fadd_rn(r, -PI_F);
}

__global__ __launch_bounds__(256) void k_reward(tb_train_io io) {
    const int S = io.n_step, A = io.n_agent;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)io.n_scene * A * S;
    if (idx >= total) return;
    const int s = (int)(idx % S);
    const int a = (int)((idx / S) % A);
    const int b = (int)(idx / ((size_t)S * A));
    const bool pv = io.pred_valid[idx] != 0;
    const float* p = io.pred_states + idx * 4;
    float reward = 0.f;
    bool rvalid = pv;
    if (io.w_collision > 0.f) {  // rewards.py:50-113
        float col = 0.f;
        if (pv) {
            const float EPS = 1.1920928955078125e-07f;  // torch.finfo(float32).eps
            const float* sz_i = io.agent_size + ((size_t)b * A + a) * 3;
            const float w_i = fminf(sz_i[0], sz_i[1]), l_i = fmaxf(sz_i[0], sz_i[1]);
            const float d_i = __fadd_rn(l_i, -w_i) / 4.0f;
            const float r_i = __fadd_rn(w_i / 2.0f, EPS);
            const float hx_i = cosf(p[2]), hy_i = sinf(p[2]);
            float cx_i[5], cy_i[5];
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                cx_i[k] = __fadd_rn(p[0], __fmul_rn(__fmul_rn((float)(k - 2), hx_i), d_i));
                cy_i[k] = __fadd_rn(p[1], __fmul_rn(__fmul_rn((float)(k - 2), hy_i), d_i));
            }
            float acc = 0.f;
            int n_valid = 0;
            for (int j = 0; j < A; ++j) {
                const size_t jdx = ((size_t)b * A + j) * S + s;
                const bool jv = io.pred_valid[jdx] != 0;
                n_valid += jv;
                if (!jv || j == a) continue;
                const float* q = io.pred_states + jdx * 4;
                const float* sz_j = io.agent_size + ((size_t)b * A + j) * 3;
                const float w_j = fminf(sz_j[0], sz_j[1]), l_j = fmaxf(sz_j[0], sz_j[1]);
                const float d_j = __fadd_rn(l_j, -w_j) / 4.0f;
                const float r_j = __fadd_rn(w_j / 2.0f, EPS);
                const float hx_j = cosf(q[2]), hy_j = sinf(q[2]);
                float dmin = INFINITY;
#pragma unroll
                for (int kj = 0; kj < 5; ++kj) {
                    const float cx = __fadd_rn(q[0], __fmul_rn(__fmul_rn((float)(kj - 2), hx_j), d_j));
                    const float cy = __fadd_rn(q[1], __fmul_rn(__fmul_rn((float)(kj - 2), hy_j), d_j));
#pragma unroll
                    for (int ki = 0; ki < 5; ++ki) {
                        const float dx = __fadd_rn(cx_i[ki], -cx), dy = __fadd_rn(cy_i[ki], -cy);
                        dmin = fminf(dmin, __fadd_rn(sqrtf(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy))), EPS));
                    }
                }
                const float c = fmaxf(__fadd_rn(1.f, -(dmin / __fadd_rn(r_j, r_i))), 0.f);
                if (io.reduce_collision_with_max) acc = fmaxf(acc, c);
                else acc = __fadd_rn(acc, fminf(c, 1.f));
            }
            col = io.reduce_collision_with_max ? acc : acc / (float)n_valid;
        }
        reward = __fadd_rn(reward, -__fmul_rn(io.w_collision, col));
    }
    if (io.use_il_loss && io.gt_valid) {  // rewards.py:115-129
        const bool gv = io.gt_valid[idx] != 0;
        const bool both = pv && gv;
        float il = 0.f;
        if (both) {
            const float* g = io.gt_states + idx * 4;
            const float e_pos = __fadd_rn(crit_eval(io.crit_pos, g[0], p[0]), crit_eval(io.crit_pos, g[1], p[1]));
            float e_rot;
            if (io.angular_type == 0) e_rot = crit_eval(io.crit_rot, g[2], p[2]);
            else if (io.angular_type == 1) e_rot = crit_eval(io.crit_rot, cast_rad(__fadd_rn(g[2], -p[2])), 0.f);
            else if (io.angular_type == 2) e_rot = __fmul_rn(0.5f, __fadd_rn(1.f, -cosf(__fadd_rn(g[2], -p[2]))));
            else e_rot = __fadd_rn(crit_eval(io.crit_rot, cosf(g[2]), cosf(p[2])), crit_eval(io.crit_rot, sinf(g[2]), sinf(p[2])));
            const float e_spd = crit_eval(io.crit_spd, g[3], p[3]);
            il = __fadd_rn(__fadd_rn(__fmul_rn(io.w_pos, e_pos), __fmul_rn(io.w_rot, e_rot)), __fmul_rn(io.w_spd, e_spd));
        } else {
            // both states are zero-filled where either is invalid: every criterion of (0, 0) is 0 except the "vector" angular
            // error, also 0 (cos 0 - cos 0); the cosine error is 0.5 (1 - cos 0) = 0
            il = 0.f;
        }
        reward = __fadd_rn(reward, -il);
        rvalid = both;
    }
    io.diffbar_rewards[idx] = rvalid ? reward : 0.f;
    io.diffbar_rewards_valid[idx] = rvalid;
}

__global__ __launch_bounds__(256) void k_train_partials(tb_train_io io, const float* post_log_std, const float* prior_log_std) {
    __shared__ double red[TP_FIELDS][4];
    const int S = io.n_step, A = io.n_agent, P = io.n_pl;
    const int row = blockIdx.x * blockDim.x + threadIdx.x;  // (scene, agent)
    float acc[TP_FIELDS];
#pragma unroll
    for (int f = 0; f < TP_FIELDS; ++f) acc[f] = 0.f;
    if (row < io.n_scene * A) {
        const size_t base = (size_t)row * S;
        bool any_pv = false;
        float r_sum = 0.f, r_cnt = 0.f;
        for (int s = 0; s < S; ++s) {
            bool pv = io.pred_valid[base + s] != 0;
            if (io.irrelevant_draw) pv = (pv && io.relevant[row] != 0) || io.irrelevant_draw[row] != 0;  // training.py:85-89
            if (!io.loss_for_teacher_forcing) pv = pv && !io.override_masks[base + s];
            if (s < io.step_training_start) pv = false;
            any_pv |= pv;
            if (pv && io.diffbar_rewards_valid[base + s]) {
                r_sum += io.diffbar_rewards[base + s];
                r_cnt += 1.f;
            }
        }
        if (io.use_diffbar_reward) {
            acc[2] = r_cnt;
            acc[3] = -r_sum;
        }
        if (io.use_vae_kl) {  // kl_divergence(posterior, prior), torch/distributions/kl.py _kl_normal_normal
            const bool kv = (io.kl_for_unseen_agent ? io.post_valid[row] : io.prior_valid[row]) && any_pv;
            if (kv) {
                float kl = 0.f;
                for (int d = 0; d < 16; ++d) {
                    const float sp = expf(post_log_std[d]), sq = expf(prior_log_std[d]);
                    const float ratio = sp / sq;
                    const float var_ratio = __fmul_rn(ratio, ratio);
                    const float t = __fadd_rn(io.post_mean[(size_t)row * 16 + d], -io.prior_mean[(size_t)row * 16 + d]) / sq;
                    const float t1 = __fmul_rn(t, t);
                    kl += __fmul_rn(0.5f, __fadd_rn(__fadd_rn(__fadd_rn(var_ratio, t1), -1.f), -logf(var_ratio)));
                }
                float err = io.kl_free_nats > 0.f ? fmaxf(kl, io.kl_free_nats) : kl;
                if (io.kl_balance_scale > 0.f)  // forward value of KL balancing (detach only changes gradients)
                    err = __fadd_rn(__fmul_rn(io.kl_balance_scale, err), __fmul_rn(__fadd_rn(1.f, -io.kl_balance_scale), err));
                acc[0] = 1.f;
                acc[1] = err;
            }
        }
        if (io.use_goal) {  // -Categorical(logits).log_prob(gt_dest)
            const bool gv = io.goal_valid[row] && any_pv;
            if (gv) {
                const float* lg = io.dest_logits + (size_t)row * P;
                float m = -INFINITY;
                for (int j = 0; j < P; ++j) m = fmaxf(m, lg[j]);
                float se = 0.f;
                for (int j = 0; j < P; ++j) se += expf(__fadd_rn(lg[j], -m));
                const int gd = io.gt_dest[row];  // (out of range: -inf logit = an impossible destination, never an out-of-bounds read)
                const float lgd = (gd >= 0 && gd < P) ? lg[gd] : -INFINITY;
                const float logp = __fadd_rn(__fadd_rn(lgd, -m), -logf(se));
                acc[4] = -logp;
                acc[5] = 1.f;
            }
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int f = 0; f < TP_FIELDS; ++f) {
        double v = (double)acc[f];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if (lane == 0) red[f][wave] = v;
    }
    __syncthreads();
    if (threadIdx.x < TP_FIELDS) {
        const double v = red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
        atomicAdd(io.out + threadIdx.x, v);
    }
}

void launch_train_partials(const tb_train_io& io, const float* post_log_std, const float* prior_log_std, hipStream_t s) {
    (void)hipMemsetAsync(io.out, 0, TP_FIELDS * sizeof(double), s);
    const size_t total = (size_t)io.n_scene * io.n_agent * io.n_step;
    hipLaunchKernelGGL(k_reward, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, io);
    const int rows = io.n_scene * io.n_agent;
    hipLaunchKernelGGL(k_train_partials, dim3((rows + 255) / 256), dim3(256), 0, s, io, post_log_std, prior_log_std);
}

}  // namespace tb
