// The thirteen "sum" states of the reference's ErrorMetrics and TrafficRuleMetrics (`src/models/metrics/logging.py:20-54, 86-129`)
// over a rollout buffer: the per-rank metric partials that the one collective of the path all-reduces (torchmetrics
// `dist_reduce_fx="sum"`).  One thread per (scene, agent, future) row walks the S steps; block reduction, then one double
// atomicAdd per field and workgroup.  HBM-bound byte / compare work.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/trafficbots_hip.h"

namespace tb {

constexpr int MP_FIELDS = 13;

__global__ __launch_bounds__(256) void k_metric_partials(tb_metric_io io) {
    __shared__ double red[MP_FIELDS][4];
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    const int n_rows = io.n_scene * io.n_agent * io.k_futures;
    float acc[MP_FIELDS];
#pragma unroll
    for (int f = 0; f < MP_FIELDS; ++f) acc[f] = 0.f;
    if (row < n_rows) {
        const int S = io.n_step;
        const int ba = row / io.k_futures;  // (scene, agent)
        const size_t base = (size_t)row * S;
        const bool relevant = io.agent_role[ba * 3] || io.agent_role[ba * 3 + 1] || io.agent_role[ba * 3 + 2];
        const bool veh = io.agent_type[ba] == 0;
        const bool tf = io.loss_for_teacher_forcing != 0;
        bool any_valid = false, a_out = false, a_col = false, a_edge = false, a_red = false, a_pas = false, a_goal = false, a_dest = false;
        for (int s = 0; s < S; ++s) {
            const bool pv = io.pred_valid[base + s] != 0, ov = io.override_masks[base + s] != 0;
            const bool keep = tf ? pv : (pv && !ov);
            if (io.gt_valid) {  // ErrorMetrics.update (:36-54)
                const bool ev = io.gt_valid[(size_t)ba * S + s] && pv && relevant && (tf || !ov);
                if (ev) {
                    const float* g = io.gt_states + ((size_t)ba * S + s) * 4;
                    const float* p = io.pred_states + (base + s) * 4;
                    const float dx = __fadd_rn(g[0], -p[0]), dy = __fadd_rn(g[1], -p[1]);
                    acc[0] += 1.f;
                    acc[1] += sqrtf(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)));
                    // cast_rad (transform_utils.py:9-11): (a + pi) % (2 pi) - pi with python's sign-of-divisor remainder
                    const float PI_F = 3.14159265358979323846f, TWO_PI_F = 6.28318530717958647692f;
                    const float t0 = __fadd_rn(__fadd_rn(g[2], -p[2]), PI_F);
                    float r = fmodf(t0, TWO_PI_F);
                    if (r != 0.f && r < 0.f) r = __fadd_rn(r, TWO_PI_F);
                    acc[2] += fabsf(__fmul_rn(__fadd_rn(r, -PI_F), 57.29577951308232f));  // torch.rad2deg
                    acc[3] += fabsf(__fadd_rn(g[3], -p[3]));
                }
            }
            // TrafficRuleMetrics.update (:100-129)
            any_valid |= tf ? pv : keep;
            const bool m = tf || keep;
            a_out |= m && io.outside_map[base + s];
            a_col |= m && io.collided[base + s];
            a_edge |= m && io.run_road_edge[base + s];
            a_red |= m && io.run_red_light[base + s];
            a_pas |= m && io.passive[base + s];
            a_goal |= m && io.goal_reached[base + s];
            a_dest |= m && io.dest_reached[base + s];
        }
        acc[4] = any_valid;
        acc[5] = any_valid && veh;
        acc[6] = a_out; acc[7] = a_col; acc[8] = a_edge; acc[9] = a_red; acc[10] = a_pas; acc[11] = a_goal; acc[12] = a_dest;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int f = 0; f < MP_FIELDS; ++f) {
        double v = (double)acc[f];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if (lane == 0) red[f][wave] = v;
    }
    __syncthreads();
    if (threadIdx.x < MP_FIELDS) {
        const double v = red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
        atomicAdd(io.out + threadIdx.x, v);
    }
}

void launch_metric_partials(const tb_metric_io& io, hipStream_t s) {
    (void)hipMemsetAsync(io.out, 0, MP_FIELDS * sizeof(double), s);
    const int n_rows = io.n_scene * io.n_agent * io.k_futures;
    hipLaunchKernelGGL(k_metric_partials, dim3((n_rows + 255) / 256), dim3(256), 0, s, io);
}

}  // namespace tb
