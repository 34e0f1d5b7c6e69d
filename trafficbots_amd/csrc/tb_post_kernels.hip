// WaymoPostProcessing as a GPU epilogue of the K-future rollout: score normalisation, selection of k_pred of
// the n_pred futures (top-k or the MTR-style greedy NMS), the MPA-style score NMS, the temperature softmax and the
// [scene, step, agent, mode] re-layout of `src/data_modules/waymo_post_processing.py:33-192`.  (`traj_aggr`, :194-295, cannot
// run in the reference -- it compares a Tensor with a python list at :231 -- and is not built.)
// One 64-thread workgroup per (scene, agent): thread p evaluates row p of the n_pred x n_pred distance matrix, lane 0 runs
// the short sequential selection, all lanes copy the selected trajectories.  Byte / compare work, HBM-bound.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/trafficbots_hip.h"

namespace tb {

constexpr int PP_MAX_PRED = 64, PP_MAX_K = 16;

__global__ __launch_bounds__(64) void k_post_process(tb_post_io io) {
    __shared__ float dist[PP_MAX_PRED][PP_MAX_PRED + 1];
    __shared__ float sc[PP_MAX_PRED];
    __shared__ int sel[PP_MAX_K];
    __shared__ float sel_s[PP_MAX_K];
    const int a = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int NP = io.n_pred, S = io.n_step, D = io.d_traj, A = io.n_agent;
    const int K = NP > io.k_pred ? io.k_pred : NP;
    // element strides of `trajs` over (scene, agent, future, step); all 0 = the contiguous [B,A,NP,S,D] layout.  A rollout buffer is
    // handed over where it lies: [B*K, A, S_all, 4] viewed as [B, A, K, S_all - s0, 4] (tb_post_io.traj_strides)
    const bool strided = io.traj_strides[0] | io.traj_strides[1] | io.traj_strides[2] | io.traj_strides[3];
    const size_t sb = strided ? (size_t)io.traj_strides[0] : (size_t)A * NP * S * D, sa = strided ? (size_t)io.traj_strides[1] : (size_t)NP * S * D;
    const size_t sp = strided ? (size_t)io.traj_strides[2] : (size_t)S * D, ss = strided ? (size_t)io.traj_strides[3] : (size_t)D;
    const float* tr = io.trajs + (size_t)b * sb + (size_t)a * sa;
    const bool need_dist = (NP > io.k_pred && io.n_mtr > 0) || io.n_mpa > 0;
    if (need_dist && tid < NP) {
        for (int j = 0; j < NP; ++j) {
            float acc = 0.f, last = 0.f;
            for (int s = io.use_ade ? 0 : S - 1; s < S; ++s) {
                const float dx = __fadd_rn(tr[j * sp + s * ss], -tr[tid * sp + s * ss]);
                const float dy = __fadd_rn(tr[j * sp + s * ss + 1], -tr[tid * sp + s * ss + 1]);
                last = sqrtf(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)));
                acc = __fadd_rn(acc, last);
            }
            dist[tid][j] = io.use_ade ? acc / (float)S : last;
        }
    }
    __syncthreads();
    if (tid == 0) {
        const int ty = io.agent_type[(size_t)b * A + a];
        float sum = 0.f;
        for (int p = 0; p < NP; ++p) sum = __fadd_rn(sum, io.scores[((size_t)b * A + a) * NP + p]);
        for (int p = 0; p < NP; ++p) sc[p] = io.scores[((size_t)b * A + a) * NP + p] / sum;  // :48
        if (NP > io.k_pred) {
            if (io.n_mtr > 0) {  // mtr_nms (:123-170)
                const float th = (ty >= 0 && ty < io.n_mtr) ? io.mtr_nms_thresh[ty] : 0.f;
                float clone[PP_MAX_PRED];
                for (int p = 0; p < NP; ++p) clone[p] = sc[p];
                for (int k = 0; k < K; ++k) {
                    int best = 0;
                    for (int p = 1; p < NP; ++p)
                        if (clone[p] > clone[best]) best = p;
                    for (int p = 0; p < NP; ++p) {
                        const float w = __fadd_rn(__fmul_rn(dist[best][p] < th ? 0.f : 1.f, 0.99f), 0.01f);
                        clone[p] = __fmul_rn(clone[p], w);
                    }
                    clone[best] = -1.f;
                    sel[k] = best;
                }
            } else {  // traj_topk (:172-192); returned in ascending mode order (the reference's order is unspecified)
                bool taken[PP_MAX_PRED];
                for (int p = 0; p < NP; ++p) taken[p] = false;
                for (int k = 0; k < K; ++k) {
                    int best = -1;
                    for (int p = 0; p < NP; ++p)
                        if (!taken[p] && (best < 0 || sc[p] > sc[best])) best = p;
                    taken[best] = true;
                }
                int k = 0;
                for (int p = 0; p < NP; ++p)
                    if (taken[p]) sel[k++] = p;
            }
            float ssum = 0.f;
            for (int k = 0; k < K; ++k) ssum = __fadd_rn(ssum, sc[sel[k]]);
            for (int k = 0; k < K; ++k) sel_s[k] = sc[sel[k]] / ssum;
        } else {
            for (int k = 0; k < K; ++k) {
                sel[k] = k;
                sel_s[k] = sc[k];
            }
        }
        if (io.n_mpa > 0) {  // mpa_nms (:83-121) on the selected modes
            if (io.valid[(size_t)b * A + a]) {
                const float th = (ty >= 0 && ty < io.n_mpa) ? io.mpa_nms_thresh[ty] : 0.f;
                int order[PP_MAX_K];
                for (int k = 0; k < K; ++k) order[k] = k;
                for (int i = 1; i < K; ++i) {  // argsort(descending) of the scores BEFORE suppression
                    const int v = order[i];
                    int j = i - 1;
                    while (j >= 0 && sel_s[order[j]] < sel_s[v]) {
                        order[j + 1] = order[j];
                        --j;
                    }
                    order[j + 1] = v;
                }
                for (int oi = 0; oi < K; ++oi) {
                    const int k = order[oi];
                    bool any = false;
                    for (int j = 0; j < K; ++j) any |= (dist[sel[k]][sel[j]] < th) && (sel_s[j] > sel_s[k]);
                    if (any) sel_s[k] = 1e-3f;
                }
            }
            float ssum = 0.f;
            for (int k = 0; k < K; ++k) ssum = __fadd_rn(ssum, sel_s[k]);
            for (int k = 0; k < K; ++k) sel_s[k] = sel_s[k] / ssum;
        }
        if (io.score_temperature > 0.f) {  // :66-67
            float z[PP_MAX_K], m = -INFINITY, esum = 0.f;
            for (int k = 0; k < K; ++k) {
                z[k] = logf(sel_s[k]) / io.score_temperature;
                m = fmaxf(m, z[k]);
            }
            for (int k = 0; k < K; ++k) {
                z[k] = expf(z[k] - m);
                esum = __fadd_rn(esum, z[k]);
            }
            for (int k = 0; k < K; ++k) sel_s[k] = z[k] / esum;
        }
        for (int k = 0; k < K; ++k) {
            io.waymo_scores[((size_t)b * A + a) * K + k] = sel_s[k];
            if (io.mode_idx) io.mode_idx[((size_t)b * A + a) * K + k] = sel[k];
        }
    }
    __syncthreads();
    // re-layout: [B,A,K,S,D] selection -> [B,S,A,K,*] (:69-80)
    const uint8_t v = io.valid[(size_t)b * A + a];
    for (int i = tid; i < S * K; i += 64) {
        const int s = i / K, k = i - s * K;
        const float* src = tr + sel[k] * sp + s * ss;
        const size_t o = (((size_t)b * S + s) * A + a) * K + k;
        io.waymo_trajs[o * 2] = src[0];
        io.waymo_trajs[o * 2 + 1] = src[1];
        if (io.waymo_yaw_bbox && D >= 3) io.waymo_yaw_bbox[o] = src[2];
        if (io.waymo_spd && D >= 4) io.waymo_spd[o] = src[3];
    }
    for (int s = tid; s < S; s += 64) io.waymo_valid[((size_t)b * S + s) * A + a] = v;
}

void launch_post_process(const tb_post_io& io, hipStream_t s) {
    hipLaunchKernelGGL(k_post_process, dim3(io.n_agent, io.n_scene), dim3(64), 0, s, io);
}

}  // namespace tb
