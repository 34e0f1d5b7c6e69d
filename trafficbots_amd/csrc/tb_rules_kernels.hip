// Flag-gated traffic-rule checks over a recorded rollout: collision (SAT on oriented boxes), road-edge
// crossing (segment intersection), red-light running and the passive-vehicle test of
// `src/utils/traffic_rule_checker.py:122-335`, evaluated on the post-override simulator states the step kernel records
// (`check_state` / `check_valid`, what the reference hands to `TrafficRuleChecker.check`, waymo_motion.py:311).  The
// checks never feed back into the simulation (only `outside_map` kills agents and only `dest_reached` gates the
// navigator, both stay in the step kernel), so they run once per rollout:
//   k_rule_step : grid (S, N), one workgroup per (instance, step)      -> the four per-step flags
//   k_rule_scan : one thread per (instance, agent), sequential over S -> sticky flags + the passive counter
// HBM-bound byte / compare work; every comparison uses the reference's operand order with un-contracted fp32 mul / add.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/trafficbots_hip.h"

namespace tb {

constexpr int RT = 256;  // threads per workgroup
constexpr int RMAX_A = 256;

__device__ __forceinline__ float mul_(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ float add_(float a, float b) { return __fadd_rn(a, b); }
__device__ __forceinline__ float sub_(float a, float b) { return __fadd_rn(a, -b); }

struct RuleP {
    int n_scene, k_rep, n_agent, n_pl, n_tl, n_hist, n_step, step_start;
    int en_collided, en_road_edge, en_red_light, en_passive;
    const float* state;      // [N,A,S,4]
    const uint8_t* valid;    // [N,A,S]
    const int32_t* agent_type;  // [B,A]
    const float* agent_size;    // [B,A,3]
    const uint8_t* map_valid;   // [B,P,20]
    const int32_t* map_type;    // [B,P]
    const float* map_pos;       // [B,P,20,2]
    const float* map_dir;       // [B,P,20,2]
    const uint8_t* tl_valid;    // [B,NH,T]
    const int32_t* tl_state;    // [B,NH,T]
    const float* tl_pos;        // [B,NH,T,2]
    uint8_t* raw;               // [4][N,A,S] per-step flags: collided, road edge, red light, passive (before the counter)
};

// ccw(A, B, C)  (traffic_rule_checker.py:595-596)
__device__ __forceinline__ bool ccw(float ax, float ay, float bx, float by, float cx, float cy) {
    return mul_(sub_(cy, ay), sub_(bx, ax)) > mul_(sub_(by, ay), sub_(cx, ax));
}

__global__ __launch_bounds__(RT) void k_rule_step(RuleP p) {
    __shared__ float box[RMAX_A][8];   // 4 corners (x, y)
    __shared__ float lin[RMAX_A][12];  // 4 edge lines (a, b, c)
    __shared__ float pose[RMAX_A][8];  // x, y, cos, sin, speed, red_len, red_wid, -
    __shared__ uint8_t flg[RMAX_A];    // bit0 valid, bit1 vehicle, bit2 pedestrian
    __shared__ uint32_t hit_edge[RMAX_A / 32], hit_lane[RMAX_A / 32];
    extern __shared__ uint8_t sep[];   // [A][A] separating-line matrix

    const int s = blockIdx.x, n = blockIdx.y, b = n / p.k_rep, tid = threadIdx.x;
    const int A = p.n_agent;
    if (tid < RMAX_A / 32) {
        hit_edge[tid] = 0;
        hit_lane[tid] = 0;
    }
    for (int a = tid; a < A; a += RT) {
        const size_t si = ((size_t)n * A + a) * p.n_step + s;
        const float4 st = *reinterpret_cast<const float4*>(p.state + si * 4);
        const float* sz = p.agent_size + ((size_t)b * A + a) * 3;
        const int ty = p.agent_type[(size_t)b * A + a];
        const float c = cosf(st.z), sn = sinf(st.z);
        // _get_agent_bbox (:518-543) on agent_size * collision_size_scale (:28-30)
        const float len = mul_(sz[0], 1.1f), wid = mul_(sz[1], 1.1f);
        const float ofx = mul_(mul_(0.5f, len), c), ofy = mul_(mul_(0.5f, len), sn);
        const float orx = mul_(mul_(0.5f, wid), sn), ory = mul_(mul_(0.5f, wid), -c);
        const float vx[4] = {add_(-ofx, orx), add_(ofx, orx), sub_(ofx, orx), sub_(-ofx, orx)};
        const float vy[4] = {add_(-ofy, ory), add_(ofy, ory), sub_(ofy, ory), sub_(-ofy, ory)};
        float bx[4], by[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            bx[q] = add_(st.x, vx[q]);
            by[q] = add_(st.y, vy[q]);
            box[a][2 * q] = bx[q];
            box[a][2 * q + 1] = by[q];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {  // line through corner q and q+1: a x + b y + c = 0 (:135-143)
            const int r = (q + 1) & 3;
            lin[a][3 * q] = sub_(by[r], by[q]);
            lin[a][3 * q + 1] = sub_(bx[q], bx[r]);
            lin[a][3 * q + 2] = sub_(mul_(bx[r], by[q]), mul_(by[r], bx[q]));
        }
        pose[a][0] = st.x; pose[a][1] = st.y; pose[a][2] = c; pose[a][3] = sn; pose[a][4] = st.w;
        pose[a][5] = mul_(mul_(sz[0], 0.5f), 0.6f);  // :68-69
        pose[a][6] = mul_(mul_(sz[1], 0.5f), 1.8f);
        flg[a] = (p.valid[si] ? 1 : 0) | (ty == 0 ? 2 : 0) | (ty == 1 ? 4 : 0);
    }
    __syncthreads();

    // ---- collision: sep[i][j] = a line of i has all four corners of j strictly on its positive side (:145-156)
    if (p.en_collided) {
        for (int pr = tid; pr < A * A; pr += RT) {
            const int i = pr / A, j = pr - i * A;
            bool any_line = false;
#pragma unroll
            for (int l = 0; l < 4; ++l) {
                bool all_pts = true;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float v = add_(add_(mul_(lin[i][3 * l], box[j][2 * q]), mul_(lin[i][3 * l + 1], box[j][2 * q + 1])),
                                         mul_(lin[i][3 * l + 2], 1.0f));
                    all_pts &= v > 0.f;
                }
                any_line |= all_pts;
            }
            sep[pr] = any_line ? 1 : 0;
        }
    }
    // ---- map nodes: road-edge crossing (:165-203) and distance to lane centres (:289-295)
    if (p.en_road_edge || p.en_passive) {
        const int n_node = p.n_pl * 20;
        for (int e0 = 0; e0 < n_node; e0 += RT) {
            const int e = e0 + tid;
            bool edge_ok = false, lane_ok = false;
            float cx = 0.f, cy = 0.f, dx = 0.f, dy = 0.f;
            if (e < n_node) {
                const size_t ei = (size_t)b * n_node + e;
                const int mt = p.map_type[(size_t)b * p.n_pl + e / 20];
                const bool mv = p.map_valid[ei] != 0;
                edge_ok = p.en_road_edge && mv && (mt == 4 || mt == 5 || mt == 7);  // :571
                lane_ok = p.en_passive && mv && mt >= 0 && mt < 3;                 // :587 (class -1 = all-zero one-hot: no type)
                cx = p.map_pos[ei * 2];
                cy = p.map_pos[ei * 2 + 1];
                dx = add_(cx, p.map_dir[ei * 2]);  // segment end = pos + dir (:575)
                dy = add_(cy, p.map_dir[ei * 2 + 1]);
            }
            if (__ballot(edge_ok || lane_ok) == 0) continue;  // (wave-uniform)
            for (int a = 0; a < A; ++a) {
                const uint8_t f = flg[a];
                bool he = false, hl = false;
                if (edge_ok && (f & 3) == 3) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int r = (q + 1) & 3;
                        const float ax = box[a][2 * q], ay = box[a][2 * q + 1], bx = box[a][2 * r], by = box[a][2 * r + 1];
                        he |= (ccw(ax, ay, cx, cy, dx, dy) != ccw(bx, by, cx, cy, dx, dy)) &&
                              (ccw(ax, ay, bx, by, cx, cy) != ccw(ax, ay, bx, by, dx, dy));
                    }
                }
                if (lane_ok) {
                    const float ux = sub_(pose[a][0], cx), uy = sub_(pose[a][1], cy);
                    hl = sqrtf(add_(mul_(ux, ux), mul_(uy, uy))) < 2.f;
                }
                if (__ballot(he) != 0 && (threadIdx.x & 63) == 0) atomicOr(&hit_edge[a >> 5], 1u << (a & 31));
                if (__ballot(hl) != 0 && (threadIdx.x & 63) == 0) atomicOr(&hit_lane[a >> 5], 1u << (a & 31));
            }
        }
    }
    __syncthreads();

    // ---- per agent: combine
    const int step = p.step_start + s;
    const int tls = min(step, p.n_hist - 1);  // :448
    const size_t plane = (size_t)p.n_scene * p.k_rep * A * p.n_step;
    for (int a = tid; a < A; a += RT) {
        const size_t oi = ((size_t)n * A + a) * p.n_step + s;
        const uint8_t f = flg[a];
        const bool va = f & 1, veh = f & 2;
        bool collided = false, red = false, passive_raw = false;
        if (p.en_collided) {
            bool free_all = true;
            for (int j = 0; j < A; ++j) {
                const uint8_t g = flg[j];
                const bool nosc = sep[a * A + j] || sep[j * A + a];
                const bool inval = (j == a) || ((f & 4) && (g & 4)) || !(va && (g & 1));  // :56-61, :159-160
                free_all &= nosc || inval;
            }
            collided = !free_all;
        }
        const float x = pose[a][0], y = pose[a][1], c = pose[a][2], sn = pose[a][3], spd = pose[a][4];
        bool red_ahead = false;
        if (p.en_red_light || p.en_passive) {
            const float x1 = add_(x, mul_(mul_(0.1f, spd), c)), y1 = add_(y, mul_(mul_(0.1f, spd), sn));
            const float rl = pose[a][5], rw = pose[a][6];
            for (int k = 0; k < p.n_tl; ++k) {
                const size_t ti = ((size_t)b * p.n_hist + tls) * p.n_tl + k;
                if (!p.tl_valid[ti]) continue;
                const int ts = p.tl_state[ti];
                const float tx = p.tl_pos[ti * 2], ty_ = p.tl_pos[ti * 2 + 1];
                if (p.en_red_light && ts == 1 && va && veh) {  // LANE_STATE_STOP (:222, :254)
                    const float d0x = sub_(tx, x), d0y = sub_(ty_, y), d1x = sub_(tx, x1), d1y = sub_(ty_, y1);
                    const bool in0 = fabsf(add_(mul_(d0x, c), mul_(d0y, sn))) < rl && fabsf(add_(mul_(d0x, sn), mul_(d0y, -c))) < rw;
                    const bool in1 = fabsf(add_(mul_(d1x, c), mul_(d1y, sn))) < rl && fabsf(add_(mul_(d1x, sn), mul_(d1y, -c))) < rw;
                    red |= in0 && !in1;
                }
                if (p.en_passive && (ts == 0 || ts == 1 || ts == 2 || ts == 4)) {  // tl_state[[0,1,2,4]].any (:308); -1 = no state set
                    const float vx = sub_(tx, x), vy = sub_(ty_, y);
                    const float nr = sqrtf(add_(mul_(vx, vx), mul_(vy, vy)));
                    red_ahead |= (nr < 10.f) && (add_(mul_(c, vx), mul_(sn, vy)) / nr > 0.95f);
                }
            }
        }
        if (p.en_passive) {
            bool ahead = false;
            for (int j = 0; j < A; ++j) {
                if (j == a || !(flg[j] & 1) || !va) continue;
                const float vx = sub_(pose[j][0], x), vy = sub_(pose[j][1], y);
                const float nr = sqrtf(add_(mul_(vx, vx), mul_(vy, vy)));
                ahead |= (nr < 10.f) && (add_(mul_(c, vx), mul_(sn, vy)) / nr > 0.95f);
            }
            const bool near = (hit_lane[a >> 5] >> (a & 31)) & 1u;
            passive_raw = va && veh && near && (spd < 5.f) && !red_ahead && !ahead;  // :328
        }
        const bool edge = p.en_road_edge && ((hit_edge[a >> 5] >> (a & 31)) & 1u) && va && veh;
        p.raw[0 * plane + oi] = collided;
        p.raw[1 * plane + oi] = edge;
        p.raw[2 * plane + oi] = red;
        p.raw[3 * plane + oi] = passive_raw;
    }
}

// sticky flags and the passive counter (:331-334, :424-465): thread per (instance, agent)
__global__ void k_rule_scan(const uint8_t* __restrict__ raw, int n_rows, int n_step, uint8_t* collided, uint8_t* collided_this,
                            uint8_t* road_edge, uint8_t* road_edge_this, uint8_t* red_light, uint8_t* red_light_this,
                            uint8_t* passive, uint8_t* passive_this, const float* __restrict__ state,
                            const uint8_t* __restrict__ valid, const float* __restrict__ goal, const float* __restrict__ agent_size,
                            int n_agent, int k_rep, uint8_t* goal_reached, uint8_t* goal_reached_this) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    const size_t plane = (size_t)n_rows * n_step, base = (size_t)r * n_step;
    bool c = false, e = false, l = false, pz = false;
    float counter = 0.f;
    if (goal && goal_reached) {  // _check_goal_reached (:337-361): within 8 agent lengths and 15 degrees of the goal pose, once
        const int n = r / n_agent, a = r % n_agent;
        const size_t ba = (size_t)(n / k_rep) * n_agent + a;
        const float gx = goal[ba * 4], gy = goal[ba * 4 + 1], gyaw = goal[ba * 4 + 2];
        const float thr_pos = mul_(agent_size[ba * 3], 8.f);
        const float thr_rot = 0.2617993877991494f;  // np.deg2rad(15)
        const float PI_F = 3.14159265358979323846f, TWO_PI_F = 6.28318530717958647692f;
        bool g = false;
        for (int s = 0; s < n_step; ++s) {
            const float* st = state + (base + s) * 4;
            const float dx = sub_(st[0], gx), dy = sub_(st[1], gy);
            const bool pos_ok = sqrtf(add_(mul_(dx, dx), mul_(dy, dy))) < thr_pos;
            float rr = fmodf(add_(sub_(st[2], gyaw), PI_F), TWO_PI_F);  // cast_rad, python remainder
            if (rr != 0.f && rr < 0.f) rr = add_(rr, TWO_PI_F);
            const bool rot_ok = fabsf(sub_(rr, PI_F)) < thr_rot;
            const bool gt_ = pos_ok && rot_ok && valid[base + s] && !g;
            g |= gt_;
            if (goal_reached_this) goal_reached_this[base + s] = gt_;
            goal_reached[base + s] = g;
        }
    }
    for (int s = 0; s < n_step; ++s) {
        const bool ct = raw[base + s], et = raw[plane + base + s], lt = raw[2 * plane + base + s], pr = raw[3 * plane + base + s];
        counter = (counter + (pr ? 1.f : 0.f)) * (pr ? 1.f : 0.f);
        const bool pt = counter > 20.f;
        c |= ct; e |= et; l |= lt; pz |= pt;
        collided_this[base + s] = ct; collided[base + s] = c;
        road_edge_this[base + s] = et; road_edge[base + s] = e;
        red_light_this[base + s] = lt; red_light[base + s] = l;
        passive_this[base + s] = pt; passive[base + s] = pz;
    }
}

int run_rule_checks(const tb_rule_io* io, int n_hist, int step_start, uint8_t* raw_ws, hipStream_t s) {
    RuleP p;
    p.n_scene = io->n_scene; p.k_rep = io->k_futures; p.n_agent = io->n_agent; p.n_pl = io->n_pl; p.n_tl = io->n_tl;
    p.n_hist = n_hist; p.n_step = io->n_step; p.step_start = step_start;
    p.en_collided = io->enable_check_collided; p.en_road_edge = io->enable_check_run_road_edge;
    p.en_red_light = io->enable_check_run_red_light; p.en_passive = io->enable_check_passive;
    p.state = io->check_state; p.valid = io->check_valid; p.agent_type = io->agent_type; p.agent_size = io->agent_size;
    p.map_valid = io->map_valid; p.map_type = io->map_type; p.map_pos = io->map_pos; p.map_dir = io->map_dir;
    p.tl_valid = io->tl_valid; p.tl_state = io->tl_state; p.tl_pos = io->tl_pos; p.raw = raw_ws;
    const int n_inst = io->n_scene * io->k_futures;
    const size_t sep_bytes = (size_t)io->n_agent * io->n_agent;
    hipLaunchKernelGGL(k_rule_step, dim3(io->n_step, n_inst), dim3(RT), sep_bytes, s, p);
    const int n_rows = n_inst * io->n_agent;
    hipLaunchKernelGGL(k_rule_scan, dim3((n_rows + 255) / 256), dim3(256), 0, s, raw_ws, n_rows, io->n_step, io->collided,
                       io->collided_this_step, io->run_road_edge, io->run_road_edge_this_step, io->run_red_light,
                       io->run_red_light_this_step, io->passive, io->passive_this_step, io->check_state, io->check_valid,
                       io->agent_goal, io->agent_size, io->n_agent, io->k_futures, io->goal_reached, io->goal_reached_this_step);
    return 0;
}

hipError_t configure_rule_kernels() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(k_rule_step), hipFuncAttributeMaxDynamicSharedMemorySize, RMAX_A * RMAX_A);
}

}  // namespace tb
