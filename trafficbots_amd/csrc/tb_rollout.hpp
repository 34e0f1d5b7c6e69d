// Kernel-side parameter blocks of the rollout path (passed by value as kernel arguments).
#pragma once
#include "tb_device.hpp"

namespace tb {

struct PolicyW {
    XLayerW as2pl[3], as2tl[3], inter[3];
    GruLayerW gru[3];
    // agent InputPeEncoder (plain row-major, VALU): W1 [32][11], W2 [32][32]
    uint32_t enc_w1, enc_b1, enc_w2, enc_b2;
    uint32_t pe_fxy;   // 12 distinct xy frequencies
    uint32_t pe_fyaw;  // 24 distinct yaw multipliers
    // add_goal: mlp_in 3x(Linear128 + LN), mlp_out Linear(256->128), Linear(128->128)
    uint32_t goal_in_w[3], goal_in_b[3], goal_in_g[3], goal_in_be[3];
    uint32_t goal_out_w1, goal_out_b1, goal_out_w2, goal_out_b2;
    // add_latent: mlp_in Linear(16->128), Linear(128->128); mlp_out as above
    uint32_t lat_in_w1, lat_in_b1, lat_in_w2, lat_in_b2;
    uint32_t lat_out_w1, lat_out_b1, lat_out_w2, lat_out_b2;
    // action head per type: Linear(128->128) packed, Linear(128->2) plain [2][128]
    uint32_t head_w1[3], head_b1[3], head_w2[3], head_b2[3];
    uint32_t head_log_std[3];  // [2] each
    uint32_t latent_log_std;   // [16]
    float max_acc[3], max_yaw_rate[3];
    float dt;
};

// offsets (in floats, into the same arena) of the XDL packings of the policy Linears: fp16 pairs or plain bf16
// (tb_device_xdl.hpp); biases / LN parameters come from PolicyW
struct XLayerX {
    uint32_t wq, wkv, wo, w1, w2;
};
struct GruLayerX {
    uint32_t wih, whh;
};
struct PolicyWX {
    XLayerX as2pl[3], as2tl[3], inter[3];
    // interaction K/V projection with norm_tgt folded in: W' = W_kv diag(gamma_tgt), b' = b_kv + W_kv beta_tgt (the three layers
    // project the SAME tile, so its normalisation (x - mean) rstd is computed once per step instead of once per layer)
    uint32_t inter_kvf[3], inter_bkvf[3];
    GruLayerX gru[3];
    uint32_t goal_out_w1, goal_out_w2, lat_out_w1, lat_out_w2;
    uint32_t head_w1[3];
};

// RolloutP::nkey_pl / nkey_tl entries (written by k_kv_hoist_x)
__host__ __device__ inline int nkey_walk(int packed) { return packed & 0xFFFF; }
__host__ __device__ inline int nkey_valid(int packed) { return packed >> 16; }

struct RolloutP {
    const float* W;  // weight arena
    PolicyW pw;
    PolicyWX px;
    // sizes
    int n_scene, k_rep, n_inst, n_agent, a_pad, n_pl, p_pad, n_tl, t_pad, n_hist, n_tl_hist, step_start, n_step_out;
    uint32_t latent_log_std;  // offset of the log_std (prior or posterior) that latent_log_prob uses
    // encoded scene
    const float* map_feature;     // [B,P,128]
    const float* tl_feature;      // [B,NH,T,128]
    // hoisted keys/values
    float* kpl;                   // [B,3,p_pad,128]
    float* vtpl;                  // [B,3,128,p_pad]
    int* nkey_pl;                 // [B]      keys the XDL step kernel walks per scene (valid polylines compacted, whole blocks) in the low
                                  //          16 bits, the exact count of valid keys in the high 16 (nkey_walk / nkey_valid)
    int* nkey_tl;                 // [B*NH]   likewise per (scene, traffic-light step)
    float* kbias_pl;              // [B,p_pad]   additive key mask: 0 valid, -inf invalid / padding
    float* ktl;                   // [B*NH,3,t_pad,128]
    float* vttl;                  // [B*NH,3,128,t_pad]
    float* kbias_tl;              // [B*NH,t_pad]
    // history / teacher forcing
    const uint8_t* hist_valid;    // [B,NH,A]
    const float* hist_state;      // [B,NH,A,4]
    const float* hist_vel;        // [B,NH,A,2]
    const float* hist_acc;        // [B,NH,A]
    const float* hist_yaw_rate;   // [B,NH,A]
    const uint8_t* tf_mask;       // [B,NH,A]
    // Per-call overrides of the stepwise API (tb_rollout_step_ex, the reference's forward(state_override=, mask_state_override=)
    // followed by Dynamics.kill(gt_valid)): when ovr_mask != nullptr the teacher-forcing inputs of THIS launch's C half come
    // from these per-INSTANCE arrays instead of step t of the per-scene history arrays above.
    const uint8_t* ovr_mask;      // [N,A]
    const float* ovr_state;       // [N,A,4]
    const float* ovr_vel;         // [N,A,2]
    const float* ovr_acc;         // [N,A]
    const float* ovr_yaw_rate;    // [N,A]
    const uint8_t* ovr_gt_valid;  // [N,A], or nullptr = no ground truth: every agent that leaves the map is killed
    // per-call action override (the reference's forward(action_override=, mask_action_override=), dynamics.py:96-100), independent of
    // the state override above: where the mask is set and the agent is valid, the physical action replaces the policy's in THIS step
    const float* ovr_action;      // [N,A,2] acc (m/s^2), yaw rate (rad/s)
    const uint8_t* ovr_action_mask;  // [N,A], or nullptr
    const int32_t* agent_type;    // [B,A]
    const float* agent_size;      // [B,A,3]
    // rule checker geometry
    const float* map_boundary;    // [B,4]
    const uint8_t* map_valid;     // [B,P,20]
    const int32_t* map_type;      // [B,P]
    const float* map_pos;         // [B,P,20,2]
    const float* map_dir;         // [B,P,20,2]
    // per instance inputs
    const float* action_eps;      // [N,A,S,2] standard-normal draws of sampled actions, or nullptr (deterministic_action)
    const float* latent_z;        // [N,A,16]  (== o_latent_z when the prologue draws it)
    const float* latent_eps;      // [N,A,16] or NULL   } latent_draw: z = det ? mean : mean + eps * exp(log_std), written to o_latent_z
    const uint8_t* latent_det;    // [N,A] or NULL      }
    float* o_latent_z;            // [N,A,16] or NULL
    int latent_draw;
    const float* latent_mean;     // [B,A,16]
    const int32_t* dest;          // [N,A]
    const uint8_t* goal_valid0;   // [N,A]
    // simulator state (workspace)
    float* state;                 // [N,a_pad,4]
    float* aux;                   // [N,a_pad,4]  vel_x, vel_y, acc, yaw_rate as last teacher-forced
    // The four arrays through which the row tiles of an instance see EACH OTHER (validity of all agents, interaction K / V) are
    // double-buffered by step parity: launch t reads what launch t-1 wrote (valid / vbias / kin / vtin = buffer t & 1) and writes
    // buffer (t + 1) & 1 (*_w), so a tile that runs late inside a launch -- second dispatch wave, another stream's kernel on the
    // chip -- never reads a sibling's step-t+1 data in its step-t interaction.  Everything else is read and written by its own tile only.
    uint8_t* valid;               // [N,a_pad]
    float* vbias;                 // [N,a_pad]  same as `valid` as additive key mask for the interaction
    uint8_t* valid_w;
    float* vbias_w;
    float* kin_w;                 // [N,3,a_pad,128]
    float* vtin_w;                // [N,3,128,a_pad]
    uint8_t* valid_b[2];          // the two buffers of each (host side: step_launch picks by parity)
    float* vbias_b[2];
    float* kin_b[2];
    float* vtin_b[2];
    uint8_t* killed;
    uint8_t* goal_valid;
    uint8_t* dest_reached;
    uint8_t* outside;
    float* hidden;                // [3,N,a_pad,128]
    float* x_mid;                 // [N,a_pad,128]  read by C(t)
    float* x_mid_w;               // written by A(t+1): the same buffer, except on the step that leaves the batched warm start
    // Batched warm start (k_step_x, A half only, grid.z = n steps): while every valid agent is teacher-forced and nobody leaves, the
    // state the A half of step t+1 starts from IS the ground truth of step t, so those A halves do not depend on the rollout and run
    // as ONE launch of n x tiles workgroups (the chip is half empty at 32 scenes); slice z of the *_pre buffers takes A(pre_t0 + z + 1)
    int pre_mode, pre_t0;
    int sw_lean_off;  // (host side only: tb_switches.step_lean resolved for this call -- the launchers pick the carve)
    float* x_mid_pre;             // [n_pre][N,a_pad,128]
    float* kin_pre;               // [n_pre][N,3,a_pad,128]
    float* vtin_pre;              // [n_pre][N,3,128,a_pad]
    // round 5: the INTERACTION of the warm-start steps is batched as well (pre_mode = 2, one more launch of n x tiles workgroups behind
    // the first): C(t + 1)'s three interaction layers read only what the batched A halves wrote (x_mid, the interaction K / V of every
    // row tile of the scene) and the ground-truth validity of step t, so slice z of x_int_pre takes the residual stream behind them,
    // once per scene; the C launches of those steps start at the GRU (skip_inter) with x_mid pointed at that slice
    float* x_int_pre;             // [n_pre][N,a_pad,128]
    float* vbias_pre;             // [n_pre][N,a_pad]   key bias of the interaction from the ground-truth validity (written by pre_mode 1)
    int skip_inter;
    float* kin;                   // [N,3,a_pad,128]
    float* vtin;                  // [N,3,128,a_pad]
    float* goal_pre;              // [N,a_pad,128]
    float* lat_pre;               // [N,a_pad,128]
    float* dest_geo;              // [N,a_pad,20,4] destination polyline nodes: px, py, unit dir x, unit dir y
    int* dest_flag;               // [N,a_pad] bit0: lane-type destination, bit1: road-edge destination
    long long* prof;              // [n_blocks][32] stage time stamps (only written by -DTB_PROFILE builds)
    // Helper workgroups (tb_device_xdl.hpp: kv_helper_x, gru_hh_helper): when a launch has at most 128 tiles, a second workgroup per
    // tile (blockIdx.z = 0: dispatched first, on a CU the launch would leave idle) computes what depends only on data of the PREVIOUS
    // launch -- the interaction K / V of layers 1, 2 from x_mid, then W_hh h of the three GRU layers from the hidden state -- and hands
    // it to the tile workgroups (blockIdx.z = 1) through L2.  nullptr: the tile workgroups compute everything themselves.
    float* gh;                    // [N * tiles][3 layers][3 gates][4 waves][2 tiles][64 lanes][4]
    unsigned int* gh_flag;        // [N * tiles]  = step + 1 once the tile's gh of that step is complete (zeroed per rollout)
    unsigned int* kv_flag;        // [N * tiles][2] = step + 1 once the tile's K / V of layer 1 / 2 of that step are stored
    unsigned int* sync_err;       // one sticky word of the context: a tile workgroup gave up waiting for a helper (tb_check_status)
    const int* warm_tab;          // L2 warmers (the helper workgroups, tb_stepx_kernels.hip): [warm_n][2] {arena offset, request time in cycles}, or nullptr
    int warm_n;
    int pre_shared;               // 1: this launch's C half reads a slice of the batched warm start, which exists once per SCENE (in the slot of
                                  // future 0: the K futures share the ground truth) -- x_mid and the interaction K / V come from instance b * K
    int dbg_helper_delay;         // test knob (env TB_DEBUG_HELPER_DELAY, clock64 ticks): the helpers start late, the hand-off runs uneven
    // outputs
    float* preds;                 // [N,A,S,4]
    uint8_t* o_valid;
    uint8_t* o_override;
    uint8_t* o_outside;
    uint8_t* o_outside_this;
    uint8_t* o_dest_reached;
    uint8_t* o_dest_reached_this;
    float* o_action_logp;         // [N,A,S]
    float* o_latent_logp;         // [N,A]
    float* o_check_state;         // [N,A,S,4] post-override state of every step (optional)
    uint8_t* o_check_valid;       // [N,A,S]
    int tap_step;                 // absolute step to tap, -1 = none, -2 = every step (the buffers hold the latest one)
    float* o_action;              // [N,A,S,2] or NULL: the physical action applied at every step
    float* tap_policy_feature;    // [N,A,128]
    float* tap_agent_feature;     // [N,A,128]
};

// reads from the buffers of parity t & 1, writes to the others
inline void set_parity(RolloutP& p, int t) {
    const int r = t & 1, w = r ^ 1;
    p.valid = p.valid_b[r]; p.vbias = p.vbias_b[r]; p.kin = p.kin_b[r]; p.vtin = p.vtin_b[r];
    p.valid_w = p.valid_b[w]; p.vbias_w = p.vbias_b[w]; p.kin_w = p.kin_b[w]; p.vtin_w = p.vtin_b[w];
}

}  // namespace tb
