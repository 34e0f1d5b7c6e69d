/*
 * trafficbots_hip.h -- C ABI of the MI355X (gfx950) scene-encoder + closed-loop rollout hot path.
 *
 * The reference (zhejz/TrafficBots) has no FFI: its seam is Python (`src/pl_modules/waymo_motion.py`,
 * `src/models/traffic_bots.py`).  Each entry point below names the reference interface it replaces;
 * `trafficbots_amd/` binds them with ctypes (see INTEGRATION.md for the stub a maintainer would add).
 *
 * Conventions
 *  - Every buffer in the *_io structs is a BORROWED DEVICE pointer (caller's allocator, e.g. PyTorch-ROCm),
 *    valid until the work queued on `stream` completes.  The library owns only its weight arena and scratch.
 *  - No hidden synchronisation: calls enqueue on `stream` and return.
 *  - Return 0 on success, non-zero on error (message via tb_last_error).  One tb_ctx per (device, stream user);
 *    not thread-safe.
 *  - Floats are fp32, masks are uint8 (0/1), indices int32, row-major, shapes in comments.
 *    B = scenes, K = futures per scene, N = B*K rollout instances (instance n uses scene n / K),
 *    A = agents, P = polylines, T = traffic-light stop points, NH = history steps (11), S = executed steps.
 */
#ifndef TRAFFICBOTS_HIP_H
#define TRAFFICBOTS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct tb_ctx tb_ctx;
typedef void* tb_stream; /* hipStream_t */

/* Launch-shaping switches of the library (round 6: these used to be environment variables only).  Every field: 0 = automatic, the
 * default and the validated choice; the other values are A/B and debugging aids -- results are bit-identical across every setting
 * except where noted.  The environment variable named beside a field still works as a DEBUGGING OVERRIDE and wins when set. */
typedef struct tb_switches {
    int32_t step_helpers;     /* helper workgroups on the idle CUs of launches of <= 128 tiles (interaction K / V, W_hh h): 0 = on while
                               *   no other context of this device has launched work in the last 25 ms (two rollouts in flight fill
                               *   the chip by themselves), 1 = off, 2 = on regardless   [TB_STEP_HELPERS=0 | 1] */
    int32_t step_l2_warmers;  /* L2 warmers on the helper CUs (+1.4 % for ONE rollout in flight, a loss when another context's work
                               *   wants those CUs): 0 = on while no other context of this device has launched work in the last
                               *   25 ms, 1 = off, 2 = on regardless   [TB_STEP_WARM=0 | 1] */
    int32_t step_pre_inter;   /* interaction blocks of the teacher-forced steps as one batched launch: 1 = off   [TB_STEP_PRE_INTER=0] */
    int32_t step_w3;          /* bf16: three-workgroups-per-CU carve for launches of > 512 tiles: 1 = off   [TB_STEP_W3=0] */
    int32_t step_aw;          /* bf16: assist-wave carve for 129..256-tile launches over >= 512 polylines (another summation order
                               *   of the same softmax: deterministic, not bit-equal to the four-wave kernel): 1 = off, 2 = forced for
                               *   launches of any size up to 256 tiles (test switch)   [TB_STEP_AW=0 | 2] */
    int32_t step_lean;        /* two-workgroups-per-CU carve for launches of > 256 tiles: 1 = off   [TB_STEP_LEAN=0] */
    int32_t rollout_graph;    /* one hipGraph per repeated rollout argument set: 1 = off (plain launches)   [TB_ROLLOUT_GRAPH=0] */
    int32_t encode_pack;      /* map encoder's polyline block: 0 = the eight-wave fused kernel with merged phases; 1..5 = the older
                               *   tilings TB_ENCODE_PACK = 0..4 (value - 1)   [TB_ENCODE_PACK] */
    int32_t encode_side;      /* agent / traffic-light tokens and the destination GRU on the library's side stream: 1 = off
                               *   [TB_ENCODE_SIDE=0] */
    int32_t encode_dest_side; /* destination predictor beside the personality branch: 1 = off   [TB_ENCODE_DEST_SIDE=0] */
    int32_t dest_lds_pad;     /* LDS padding (bytes) of the destination-pair kernel when it runs beside the personality branch:
                               *   0 = 30000, -1 = none   [TB_DEST_LDS_PAD] */
    int32_t reserved[5];      /* must be 0 */
} tb_switches;

/* Scalar config the kernels need (defaults = configs/model/traffic_bots.yaml of the reference). */
typedef struct tb_config {
    int32_t time_step_current;   /* 10  (traffic_bots.yaml:5)  */
    int32_t time_step_sim_start; /* 1   (traffic_bots.yaml:8)  */
    float dt;                    /* 0.1 (src/utils/dynamics.py:13) */
    float max_acc[3];            /* per type veh,ped,cyc (traffic_bots.yaml:142-155) */
    float max_yaw_rate[3];
    float action_log_std;        /* -2 (traffic_bots.yaml:138) */
    float latent_log_std;        /* -1 (traffic_bots.yaml:76); overwritten by the loaded parameter */
    int32_t operand_precision;   /* 0 (default): fp32-accurate (fp16-pair MFMA operands, fp32 accumulate), with an AUTOMATIC fallback to
                                  *    the exact-fp32 kernels when a loaded tensor or a run-time activation leaves the fp16-pair range
                                  *    (|x| < 65504): tb_finalize_weights / tb_check_status / tb_precision_state;
                                  * 1: bf16 MFMA operands, fp32 accumulate -- BASELINE.json configs 4/5, no fp32-parity claim;
                                  * 2: exact fp32 from the start (fp32 MFMA kernels: fp32's range, about 2x slower per step) */
    tb_switches sw;              /* launch-shaping switches; all zero = automatic */
} tb_config;

/* -- lifecycle ----------------------------------------------------------------------------------- */
/* Replaces: WaymoMotion.__init__ / hydra.utils.instantiate (waymo_motion.py:28-106). */
int tb_create(const tb_config* cfg, tb_ctx** out);
void tb_destroy(tb_ctx* ctx);
const char* tb_last_error(tb_ctx* ctx);
/* Range check of the fp32-accurate mode (the one call of this header that SYNCHRONISES `stream`).  The XDL kernels carry an fp32
 * value as an fp16 pair, valid for |x| < 65504; a GEMM / attention operand beyond that raises a sticky device flag where it is
 * produced (the overflow would otherwise turn into inf / NaN that ReLU and the softmax clamp can squash silently).  Returns 0 when
 * no kernel queued on this device since the previous check raised it; the flag is cleared either way.
 * Returns 3 when it was raised: the results of the calls since the previous check are invalid AND the context has switched itself
 * to the exact-fp32 twins of the kernels that overflowed (fp32's range) -- re-issue those calls (the reference has no range limit,
 * src/models/modules/mlp.py:20-85, so neither has this library: it only gets slower).  Any other non-zero value is a hard error
 * (tb_last_error).  operand_precision = 1 (bf16) and 2 (exact fp32) have fp32's range and never raise the flag. */
int tb_check_status(tb_ctx* ctx, tb_stream stream);
/* Which kernels the context runs on: out[0] = step kernels, out[1] = scene encoders (0 fp16-pair XDL, 1 bf16, 2 exact fp32 MFMA);
 * out[2] = why it left the configured ones (bit 0: a loaded tensor is outside the fp16-pair range, seen by tb_finalize_weights;
 * bit 1: a run-time activation overflow, seen by tb_check_status).  tb_precision_note: the sentence naming the tensor / stage. */
int tb_precision_state(tb_ctx* ctx, int32_t out[3]);
const char* tb_precision_note(tb_ctx* ctx);
/* Back to the kernels tb_finalize_weights selected after a RUN-TIME fallback (tb_check_status returned 3; reason bit 1): both
 * operand packings of every weight live in the arena, so this is a state change, no upload and no synchronisation.  For the caller
 * whose overflow was ONE outlier batch: re-issue that batch on the exact-fp32 kernels, then call this and continue at full speed (the
 * host mirror does it: WaymoMotion.fallback_policy).  A context whose LOADED tensors are outside the range (reason bit 0) stays on
 * the exact kernels; a context that has not fallen back is left alone.  Closes an open stepwise rollout (tb_rollout_begin again).
 * Returns 0; out_changed (may be NULL) = 1 when the kernel selection changed. */
int tb_precision_restore(tb_ctx* ctx, int32_t* out_changed);
/* Version / build info string (static storage). */
const char* tb_version(void);

/* -- weights ------------------------------------------------------------------------------------- */
/* Replaces: LightningModule.load_from_checkpoint / load_state_dict (src/run.py:42-44).
 * `name` is the reference state_dict key, `data` a HOST fp32 array of `numel`.
 * After all tensors are staged, tb_finalize_weights packs them into MFMA fragment order and uploads. */
int tb_load_weight(tb_ctx* ctx, const char* name, const float* host_data, int64_t numel);
int tb_finalize_weights(tb_ctx* ctx, tb_stream stream);

/* -- hot path: closed-loop rollout ------------------------------------------------------------------ */
typedef struct tb_rollout_io {
    /* sizes */
    int32_t n_scene, k_futures, n_agent, n_pl, n_tl, n_hist, step_end;
    /* encoded scene (outputs of encode_input_features, traffic_bots.py:109-151) */
    const float* map_feature;      /* [B,P,128] */
    const uint8_t* map_feature_valid; /* [B,P] */
    const float* tl_feature;       /* [B,NT,T,128]  NT = n_tl_step ? n_tl_step : NH */
    const uint8_t* tl_feature_valid;  /* [B,NT,T] */
    /* history (test_step: NH = time_step_current+1) or full ground truth (validation: NH = 91) used for init / teacher
     * forcing / the kill rule (features{} of joint_future_pred / reactive_replay, waymo_motion.py:533-546, 448-462) */
    const uint8_t* agent_valid;    /* [B,NH,A] */
    const float* agent_state;      /* [B,NH,A,4] x,y,yaw,spd */
    const float* agent_vel;        /* [B,NH,A,2] */
    const float* agent_acc;        /* [B,NH,A] */
    const float* agent_yaw_rate;   /* [B,NH,A] */
    const uint8_t* mask_teacher_forcing; /* [B,NH,A] (TeacherForcing.get, teacher_forcing.py:33-74) */
    const int32_t* agent_type;     /* [B,A] 0 veh / 1 ped / 2 cyc / -1 none */
    const float* agent_size;       /* [B,A,3] */
    /* rule-checker geometry (TrafficRuleChecker.__init__, traffic_rule_checker.py:77-98) */
    const float* map_boundary;     /* [B,4] xmin,xmax,ymin,ymax */
    const uint8_t* map_valid;      /* [B,P,20] */
    const int32_t* map_type;       /* [B,P] index of the one-hot */
    const float* map_pos;          /* [B,P,20,2] */
    const float* map_dir;          /* [B,P,20,2] */
    /* per instance */
    const float* latent_sample;    /* [N,A,16] z (MyDist.sample, distributions.py:18-38) */
    const float* latent_mean;      /* [B,A,16] prior mean, for latent_log_prob */
    const int32_t* dest;           /* [N,A] destination polyline index (goal_sample) */
    const uint8_t* goal_valid;     /* [N,A] */
    /* outputs = RolloutBuffer fields after finish() (buffer.py:72-90), step axis S = step_end - sim_start + 1 */
    float* preds;                  /* [N,A,S,4] */
    uint8_t* valid;                /* [N,A,S] */
    uint8_t* override_masks;       /* [N,A,S] */
    uint8_t* outside_map;          /* [N,A,S] */
    uint8_t* outside_map_this_step;/* [N,A,S] */
    uint8_t* dest_reached;         /* [N,A,S] */
    uint8_t* dest_reached_this_step; /* [N,A,S] */
    float* action_log_probs;       /* [N,A,S] */
    float* latent_log_prob;        /* [N,A]   (constant over steps, traffic_bots.py:196-199) */
    /* final simulator state (Dynamics.agent_state/agent_valid, TrafficBots.hidden) -- may be NULL */
    float* final_state;            /* [N,A,4] */
    uint8_t* final_valid;          /* [N,A] */
    float* final_hidden;           /* [3,N,A,128] */
    /* optional debug taps (NULL to skip): policy feature of one step */
    int32_t tap_step;              /* absolute step whose policy feature to capture, -1 = none, -2 = every step (the buffers then hold
                                    * the latest: after tb_rollout_begin / each tb_rollout_step, tap_agent_feature is the agent feature
                                    * the NEXT step's policy reads -- what a stepwise caller needs to ask tb_forward for that step's
                                    * attention weights, require_vis_dict of waymo_motion.py:167,191-201) */
    float* tap_policy_feature;     /* [N,A,128] */
    float* tap_agent_feature;      /* [N,A,128] */
    /* optional (NULL to skip): the post-override simulator state of every step, i.e. what the reference passes to
     * TrafficRuleChecker.check (waymo_motion.py:311) -- input of tb_rule_checks */
    float* check_state;            /* [N,A,S,4] */
    uint8_t* check_valid;          /* [N,A,S]   */
    /* number of steps held by tl_feature / tl_feature_valid when it differs from n_hist (0 = n_hist).  reactive_replay
     * (waymo_motion.py:448-462) overrides from the 91-step ground truth (n_hist = 91) while the traffic-light features are
     * those of the 11 history steps. */
    int32_t n_tl_step;
    /* nonzero: latent_sample / latent_mean come from the POSTERIOR (reactive_replay, waymo_motion.py:597-611), so
     * latent_log_prob uses latent_post_dist.log_std instead of latent_prior_dist.log_std */
    int32_t latent_posterior;
    /* W > 0: the caller guarantees that for every step t <= W (t < n_hist) each agent that is valid in agent_valid[:, t] is
     * teacher-forced (mask_teacher_forcing[:, t] set) and that no agent turns from valid to invalid within steps 0..W (the
     * default warm start, TeacherForcing.step_warm_start = time_step_current, on histories without early exits).  Then the
     * simulator state after each of those steps IS the ground truth of that step and the map / traffic-light attention half of
     * the steps up to W + 1 does not depend on the rollout: it is run as one batched launch that fills the chip.  Results are
     * bit-identical to W = 0, which is always allowed. */
    int32_t warm_start_steps;
    /* optional (NULL = deterministic actions): standard-normal draws for `deterministic_action=False` (Dynamics.update,
     * dynamics.py:77 -> MyDist.sample -> Normal.rsample, distributions.py:18-38): step s samples the unbounded action as
     * mean + action_eps[:, :, s] * exp(log_std) and action_log_probs becomes Normal.log_prob of the sample (summed over the two
     * action dims).  XDL step kernels only. */
    const float* action_eps;       /* [N,A,S,2] */
    /* optional (NULL = never): train-mode `p_drop_hidden` (waymo_motion.py:345-351) with the caller's draws -- a HOST array of S
     * bytes, read while the launches are enqueued: where hidden_drop[s] is set the GRU hidden state of ALL instances is zeroed after
     * step s has been recorded (the reference draws one `torch.rand(1) < p_drop_hidden` per step for the whole batch).
     * tb_rollout ONLY: tb_rollout_begin fails when it is set (a stepwise caller zeroes the state between its own step calls). */
    const uint8_t* hidden_drop;    /* [S] host memory */
    /* optional (round 4): draw the personalities INSIDE the rollout prologue (`MyDist.sample`, distributions.py:18-38, called by
     * TrafficBots.forward on the first step, traffic_bots.py:196-199).  When latent_sample_out is non-NULL the prologue computes
     *   z[n,a,:] = deterministic[n,a] ? latent_mean[n / K, a, :] : latent_mean[n / K, a, :] + latent_eps[n,a,:] * exp(log_std)
     * (log_std of the prior, or of the posterior with latent_posterior), writes it to latent_sample_out, and uses it as the sample;
     * `latent_sample` is then ignored and may be NULL.  latent_eps NULL = every agent takes the mean; latent_deterministic NULL =
     * no agent is deterministic. */
    const float* latent_eps;             /* [N,A,16] standard-normal draws, or NULL */
    const uint8_t* latent_deterministic; /* [N,A] or NULL */
    float* latent_sample_out;            /* [N,A,16] or NULL */
    /* optional (NULL to skip): the physical action (acceleration m/s^2, yaw rate rad/s) applied at every step -- after the tanh bound
     * and after an action override, 0 for agents that are invalid before the step: `vis_dict["action"]` of WaymoMotion.forward
     * (waymo_motion.py:191-194; Dynamics.update, dynamics.py:82-100) */
    float* actions;                      /* [N,A,S,2] */
} tb_rollout_io;

/* Replaces: WaymoMotion.rollout (+ per-step WaymoMotion.forward, TrafficBots.forward, ActionHead,
 * Dynamics.update/override_states/kill, TrafficRuleChecker.check, GoalManager.get_goal_feature /
 * disable_goal_reached, RolloutBuffer) -- waymo_motion.py:108-354. */
int tb_rollout(tb_ctx* ctx, const tb_rollout_io* io, tb_stream stream);

/* Stepwise form of the same rollout, for callers that drive the loop themselves the way the reference's rollout() drives
 * the stateful WaymoMotion.forward (waymo_motion.py:108-203, 269-343):
 *   tb_rollout_begin : prologue (K/V hoists, simulator init from history frame 0) and the first half-step; `io` as above
 *                      (its buffers must stay valid until the last step has run; step_end bounds the number of steps);
 *   tb_rollout_step  : one simulation step t = sim_start, sim_start+1, ...: writes slot t - sim_start of every output;
 *   tb_rollout_state : copy of the CURRENT simulator state (Dynamics.agent_state / agent_valid, TrafficBots.hidden);
 *                      any pointer may be NULL.
 * tb_rollout(io) == tb_rollout_begin(io) + (step_end - sim_start + 1) x tb_rollout_step, fused into fewer launches. */
int tb_rollout_begin(tb_ctx* ctx, const tb_rollout_io* io, tb_stream stream);
int tb_rollout_step(tb_ctx* ctx, tb_stream stream);

/* Per-call teacher forcing of ONE step, the reference's `forward(..., state_override=, mask_state_override=)` followed by
 * `Dynamics.kill(violations, gt_valid)` (waymo_motion.py:112-119,177,269-314; dynamics.py:132-167): where mask[n,a] is set (and the
 * agent was not killed before) the agent becomes valid and takes agent_state / vel / acc / yaw_rate -- this is also how agents are
 * spawned -- and an agent that leaves the map is killed unless gt_valid[n,a] is set (gt_valid NULL: killed always).  All arrays are
 * per INSTANCE (N = n_scene * k_futures), device pointers borrowed until the stream work completes.  The history arrays and the
 * teacher-forcing mask given to tb_rollout_begin are NOT consulted for this step. */
typedef struct tb_step_override {
    const uint8_t* mask;         /* [N,A]   mask_state_override */
    const float* agent_state;    /* [N,A,4] state_override["agent_state"]: x, y, yaw, spd */
    const float* vel;            /* [N,A,2] state_override["vel"] */
    const float* acc;            /* [N,A]   state_override["acc"] */
    const float* yaw_rate;       /* [N,A]   state_override["yaw_rate"] */
    const uint8_t* gt_valid;     /* [N,A] or NULL */
    /* `forward(action_override=, mask_action_override=)` (waymo_motion.py:116-117,174-175 -> Dynamics.update, dynamics.py:96-100): where
     * action_mask[n,a] is set and the agent is valid BEFORE this step, the dynamics update of this step uses action[n,a] = (acceleration
     * m/s^2, yaw rate rad/s) instead of the policy's tanh-bounded action; action_log_prob stays that of the policy's own action.  Both
     * or neither.  Independent of the state override: with mask == NULL (and the four state arrays NULL) the step takes the teacher
     * forcing bound at tb_rollout_begin, as tb_rollout_step does. */
    const float* action;         /* [N,A,2] or NULL */
    const uint8_t* action_mask;  /* [N,A]   or NULL */
} tb_step_override;
/* tb_rollout_step_ex(ctx, NULL, s) == tb_rollout_step(ctx, s) */
int tb_rollout_step_ex(tb_ctx* ctx, const tb_step_override* ov, tb_stream stream);
int tb_rollout_state(tb_ctx* ctx, float* state /*[N,A,4]*/, uint8_t* valid /*[N,A]*/, float* hidden /*[3,N,A,128]*/, tb_stream stream);

/* -- hot path: scene encoders ------------------------------------------------------------------------ */
typedef struct tb_encode_io {
    int32_t n_scene, n_agent, n_pl, n_tl, n_hist;
    /* raw scene (packed-h5 test split, data_h5_womd.py:119-157; bool tensors as uint8) */
    const uint8_t* agent_valid;    /* [B,NH,A] */
    const float* agent_pos;        /* [B,NH,A,2] */
    const float* agent_yaw;        /* [B,NH,A] */
    const float* agent_vel;        /* [B,NH,A,2] */
    const float* agent_spd;        /* [B,NH,A] */
    const float* agent_acc;        /* [B,NH,A] */
    const float* agent_yaw_rate;   /* [B,NH,A] */
    const int32_t* agent_type;     /* [B,A] */
    const float* agent_size;       /* [B,A,3] */
    const uint8_t* map_valid;      /* [B,P,20] */
    const int32_t* map_type;       /* [B,P] */
    const float* map_pos;          /* [B,P,20,2] */
    const float* map_dir;          /* [B,P,20,2] */
    const uint8_t* tl_valid;       /* [B,NH,T] */
    const int32_t* tl_state;       /* [B,NH,T] index of the one-hot */
    const float* tl_pos;           /* [B,NH,T,2] */
    const float* tl_dir;           /* [B,NH,T,2] */
    /* outputs */
    float* map_feature;            /* [B,P,128] */
    uint8_t* map_feature_valid;    /* [B,P] */
    float* agent_feature;          /* [B,NH,A,128] */
    float* tl_feature;             /* [B,NH,T,128] */
    float* latent_mean;            /* [B,A,16]  prior mean (LatentEncoder.forward, latent_encoder.py:70-147) */
    uint8_t* latent_valid;         /* [B,A] */
    float* dest_logits;            /* [B,A,P] masked, un-normalised (DestPredictor.forward, goal_manager.py:202-333) */
    /* Optional: the reference's OWN encoder inputs -- `TrafficBots.encode_input_features(agent_attr, agent_pe, map_attr, map_pe,
     * tl_attr, tl_pe, ...)` as `SceneCentricInput` produced them (traffic_bots.py:109-151, sc_input.py:107-140): attribute vectors
     * (agent 11 = vel2 spd yaw_rate acc size3 type3; map 31 = type11 node-one-hot20; tl 5 = state one-hot) and the 96-wide pose PE,
     * fp32, taken AS GIVEN instead of being assembled / evaluated from the raw fields above.  Per token kind both or neither; when a
     * pair is given, that kind's raw position / yaw / velocity ... fields above may be NULL (the class indices agent_type / map_type
     * and all *_valid stay required: the destination predictor and the masks read them). */
    const float* ext_agent_attr;   /* [B,NH,A,11] or NULL */
    const float* ext_agent_pe;     /* [B,NH,A,96] */
    const float* ext_map_attr;     /* [B,P,20,31] or NULL */
    const float* ext_map_pe;       /* [B,P,20,96] */
    const float* ext_tl_attr;      /* [B,NH,T,5] or NULL */
    const float* ext_tl_pe;        /* [B,NH,T,96] */
} tb_encode_io;

/* Replaces: SceneCentricInput.forward (sc_input.py:50-140) + TrafficBots.encode_input_features
 * (traffic_bots.py:109-151) + LatentEncoder.forward prior + GoalManager.pred_goal. */
int tb_encode_scene(tb_ctx* ctx, const tb_encode_io* io, tb_stream stream);

/* Posterior personality over a full ground-truth episode.
 * Replaces: SceneCentricLatent.forward the latent_post inputs (sc_latent.py:150-163,196-217, eval mode) + the agent / traffic-light
 * part of TrafficBots.encode_input_features + LatentEncoder.forward(posterior=True) (latent_encoder.py:98-136) as called by
 * validation_step / training_step (waymo_motion.py:583,597 / 367,382).  map_feature is tb_encode_scene's output for the
 * same scenes (the reference encodes the identical map inputs a second time). */
typedef struct tb_posterior_io {
    int32_t n_scene, n_agent, n_pl, n_tl, n_step;   /* n_step = ground-truth length (91), (n_step-1) % 5 == 0 */
    const uint8_t* agent_valid;    /* [B,NS,A] */
    const float* agent_pos;        /* [B,NS,A,2] */
    const float* agent_yaw;        /* [B,NS,A] */
    const float* agent_vel;        /* [B,NS,A,2] */
    const float* agent_spd;        /* [B,NS,A] */
    const float* agent_acc;        /* [B,NS,A] */
    const float* agent_yaw_rate;   /* [B,NS,A] */
    const int32_t* agent_type;     /* [B,A] */
    const float* agent_size;       /* [B,A,3] */
    const uint8_t* tl_valid;       /* [B,NS,T] */
    const int32_t* tl_state;       /* [B,NS,T] */
    const float* tl_pos;           /* [B,NS,T,2] */
    const float* tl_dir;           /* [B,NS,T,2] */
    const float* map_feature;      /* [B,P,128] */
    const uint8_t* map_feature_valid; /* [B,P] */
    float* latent_mean;            /* [B,A,16] posterior mean */
    uint8_t* latent_valid;         /* [B,A] */
} tb_posterior_io;
int tb_encode_posterior(tb_ctx* ctx, const tb_posterior_io* io, tb_stream stream);

/* Forward training losses of a replayed episode: the per-step differentiable reward and the six "sum" states of
 * TrainingMetrics.
 * Replaces: DifferentiableReward.get (src/utils/rewards.py:33-131; config group differentiable_reward,
 * configs/model/traffic_bots.yaml:157-171) as called once per step by WaymoMotion.rollout (waymo_motion.py:320-330) -- it is
 * a pure function of that step's prediction and ground truth, so it is evaluated over the recorded buffer -- AngularError
 * (src/models/metrics/loss.py:9-32), BalancedKL forward value (:35-74) and TrainingMetrics.update
 * (src/models/metrics/training.py:62-139; group training_metrics, traffic_bots.yaml:208-219).  K = 1 (reactive_replay).
 * out[6], in this order: vae_kl_counter, vae_kl, diffbar_reward_counter, diffbar_reward, goal_loss, goal_counter -- the
 * states a multi-GPU run SUM-all-reduces; TrainingMetrics.compute() forms the ratios.  Not built: w_relevant_agent > 0 (the
 * reference's own weighting there does not broadcast, training.py:124).  Criteria: 0 SmoothL1Loss, 1 MSELoss, 2 L1Loss; angular_type: 0 null, 1 cast, 2 cosine, 3 vector. */
typedef struct tb_train_io {
    int32_t n_scene, n_agent, n_step, n_pl;
    /* differentiable_reward */
    float w_collision;
    int32_t reduce_collision_with_max, use_il_loss;
    int32_t crit_pos, crit_rot, angular_type, crit_spd;
    float w_pos, w_rot, w_spd;
    /* training_metrics */
    int32_t use_vae_kl, use_diffbar_reward, use_goal;   /* w_* > 0 */
    int32_t kl_for_unseen_agent, loss_for_teacher_forcing, step_training_start;
    float kl_balance_scale, kl_free_nats;
    /* rollout buffer (un-flattened, K = 1) */
    const uint8_t* pred_valid;      /* [B,A,S] */
    const float* pred_states;       /* [B,A,S,4] */
    const uint8_t* override_masks;  /* [B,A,S] */
    const uint8_t* gt_valid;        /* [B,A,S] or NULL (no ground truth for these steps: no imitation term) */
    const float* gt_states;         /* [B,A,S,4] or NULL */
    const float* agent_size;        /* [B,A,3] */
    /* destination prediction (may be NULL when use_goal == 0) */
    const float* dest_logits;       /* [B,A,P] masked, un-normalised (tb_encode_io.dest_logits) */
    const uint8_t* goal_valid;      /* [B,A]  goal_pred.valid */
    const int32_t* gt_dest;         /* [B,A] */
    /* personalities (may be NULL when use_vae_kl == 0) */
    const float* post_mean;         /* [B,A,16] */
    const uint8_t* post_valid;      /* [B,A] */
    const float* prior_mean;        /* [B,A,16] */
    const uint8_t* prior_valid;     /* [B,A] */
    /* outputs */
    float* diffbar_rewards;         /* [B,A,S] */
    uint8_t* diffbar_rewards_valid; /* [B,A,S] */
    double* out;                    /* [6] device buffer */
    /* optional, both or neither (p_loss_for_irrelevant > 0, training.py:85-89): pred_valid := (pred_valid & relevant) |
     * irrelevant_draw, broadcast over the steps; relevant = agent_role.any(-1), irrelevant_draw = the caller's Bernoulli(p) draw */
    const uint8_t* relevant;        /* [B,A] */
    const uint8_t* irrelevant_draw; /* [B,A] */
} tb_train_io;
int tb_train_partials(tb_ctx* ctx, const tb_train_io* io, tb_stream stream);

/* -- the policy trunk as a stand-alone call, with attention weights ------------------------------------ */
/* Replaces: TrafficBots.forward(agent_valid, agent_feature, map_valid, map_feature, tl_valid, tl_feature, goal_valid, goal_feature,
 * need_weights) -> (policy_feature, latent_logp, attn_pl, attn_tl, attn_agent)  (src/models/traffic_bots.py:163-247) for the default
 * configuration (interaction_first, add_goal_latent_first = False, no final MLP): 3 x agent->map, 3 x agent->traffic-light,
 * 3 x agent<->agent attention (eye mask, single-agent bypass), one step of the 3-layer GRU, add_goal, add_latent.  With
 * need_weights the reference returns the head-mean attention weights of the LAST layer of each block (attention.py:115-146,
 * transformer.py:82-95, agent_interaction.py:61-93; rows without an admissible key and single-agent instances are zero) -- the
 * attn_* pointers (NULL = not wanted).  The fused step kernel never materialises them; this call is the un-fused visualisation /
 * debugging path (plain fp32 kernels over row-major copies of the weights, ~70 launches), and an on-device cross-check of the fused
 * kernel.  `hidden` is read and overwritten (TrafficBots.hidden; zeros = the reference's `hidden = None`).  `latent_sample` is the
 * personality drawn at TrafficBots.init time (latent_logp is the caller's: DiagGaussian.log_prob).  goal_feature / goal_valid NULL:
 * no goal (add_goal is skipped, as the reference does for goal_feature = None).  All tensors per INSTANCE, device pointers. */
typedef struct tb_forward_io {
    int32_t n_inst, n_agent, n_pl, n_tl;
    const uint8_t* agent_valid;    /* [N,A] */
    const float* agent_feature;    /* [N,A,128] */
    const uint8_t* map_valid;      /* [N,P] */
    const float* map_feature;      /* [N,P,128] */
    const uint8_t* tl_valid;       /* [N,T] */
    const float* tl_feature;       /* [N,T,128] */
    const uint8_t* goal_valid;     /* [N,A] or NULL */
    const float* goal_feature;     /* [N,A,128] or NULL */
    const float* latent_sample;    /* [N,A,16] */
    float* hidden;                 /* [3,N,A,128] in / out */
    float* policy_feature;         /* [N,A,128] out */
    float* attn_pl;                /* [N,A,P] out or NULL */
    float* attn_tl;                /* [N,A,T] out or NULL */
    float* attn_agent;             /* [N,A,A] out or NULL */
} tb_forward_io;
int tb_forward(tb_ctx* ctx, const tb_forward_io* io, tb_stream stream);

/* -- instrumentation ---------------------------------------------------------------------------------- */
/* Per-launch durations (ms, HIP events recorded on `stream`) of the LAST tb_rollout when timing was enabled with
 * tb_set_timing(ctx, 1).  A rollout of S steps issues S+1 step launches: A(1) alone, S-1 fused launches
 * C(t)+A(t+1), C(S) alone.  out[0] = sum over the fused launches, out[1] = the two edge launches, out[2] = prologue
 * (K/V hoists + init), out[3] = number of fused launches.  Timing inserts event records only (no host sync until
 * this query, which synchronises on the last event). */
int tb_set_timing(tb_ctx* ctx, int enable);
/* tb_rollout captures ONE hipGraph per rollout the second time it sees the same argument set (buffers, sizes, switches) and replays
 * it afterwards (TB_ROLLOUT_GRAPH=0: never): out2[0] = rollouts captured, out2[1] = rollouts replayed, since tb_create. */
int tb_graph_stats(tb_ctx* ctx, int32_t* out2);
int tb_get_timing(tb_ctx* ctx, float* out4);

/* Flag-gated traffic-rule checks over a recorded rollout.
 * Replaces: TrafficRuleChecker._check_collided / _check_run_road_edge / _check_run_red_light / _check_passive and their
 * accumulation in TrafficRuleChecker.check (src/utils/traffic_rule_checker.py:122-335, 412-516) for
 * traffic_rule_checker.enable_check_* = True (configs/model/traffic_bots.yaml:240-244).  The checks are functions of the
 * per-step state only and never feed back into the simulation, so they are evaluated once per rollout on
 * tb_rollout_io.check_state / check_valid.  A disabled check returns zeros in both of its arrays, as the reference does.
 * All outputs [N,A,S] uint8; scene tensors un-repeated ([B, ...], instance n -> scene n / K). */
typedef struct tb_rule_io {
    int32_t n_scene, k_futures, n_agent, n_pl, n_tl, n_step;
    int32_t enable_check_collided, enable_check_run_road_edge, enable_check_run_red_light, enable_check_passive;
    const float* check_state;      /* [N,A,S,4] */
    const uint8_t* check_valid;    /* [N,A,S]   */
    const int32_t* agent_type;     /* [B,A]  0 veh, 1 ped, 2 cyc */
    const float* agent_size;       /* [B,A,3] */
    const uint8_t* map_valid;      /* [B,P,20] */
    const int32_t* map_type;       /* [B,P] */
    const float* map_pos;          /* [B,P,20,2] */
    const float* map_dir;          /* [B,P,20,2] */
    const uint8_t* tl_valid;       /* [B,NH,T] */
    const int32_t* tl_state;       /* [B,NH,T]  0 unknown, 1 stop, 2 caution, 3 go, 4 flashing */
    const float* tl_pos;           /* [B,NH,T,2] */
    uint8_t* collided;
    uint8_t* collided_this_step;
    uint8_t* run_road_edge;
    uint8_t* run_road_edge_this_step;
    uint8_t* run_red_light;
    uint8_t* run_red_light_this_step;
    uint8_t* passive;
    uint8_t* passive_this_step;
    /* steps held by tl_valid / tl_state / tl_pos (0 = time_step_current + 1: the history, as joint_future_pred passes;
     * reactive_replay passes the 91-step "tl_stop/..." arrays, waymo_motion.py:433-447) */
    int32_t n_tl_step;
    /* optional: ground-truth goal pose [B,A,4] (x,y,yaw,spd) and the goal-reached flags of
     * TrafficRuleChecker._check_goal_reached (:337-361); the reference evaluates them whenever the batch carries
     * "agent/goal" (validation / training).  NULL to skip. */
    const float* agent_goal;
    uint8_t* goal_reached;
    uint8_t* goal_reached_this_step;
} tb_rule_io;
int tb_rule_checks(tb_ctx* ctx, const tb_rule_io* io, tb_stream stream);

/* Post-processing of the K joint futures into the k_pred scored modes of a Waymo motion submission.
 * Replaces: WaymoPostProcessing.forward / traj_topk / mtr_nms / mpa_nms (src/data_modules/waymo_post_processing.py:33-192;
 * config group waymo_post_processing, configs/model/traffic_bots.yaml:179-186).  traj_aggr (aggr_thresh != []) is not built:
 * the reference raises TypeError there (:231).  n_pred <= 64, k_pred <= 16. */
typedef struct tb_post_io {
    int32_t n_scene, n_agent, n_pred, n_step, d_traj; /* d_traj in 2..4: x, y[, yaw[, speed]] */
    int32_t k_pred;
    float score_temperature;                          /* <= 0: off */
    int32_t n_mpa;                                    /* 0: off, else 3 */
    float mpa_nms_thresh[3];                          /* veh, ped, cyc [m] */
    int32_t n_mtr;
    float mtr_nms_thresh[3];
    int32_t use_ade;
    const uint8_t* valid;       /* [B,A] */
    const float* scores;        /* [B,A,NP] un-normalised */
    const float* trajs;         /* [B,A,NP,S,D] */
    const int32_t* agent_type;  /* [B,A] */
    float* waymo_trajs;         /* [B,S,A,K,2]   K = min(k_pred, n_pred) */
    float* waymo_yaw_bbox;      /* [B,S,A,K,1] or NULL */
    float* waymo_spd;           /* [B,S,A,K,1] or NULL */
    float* waymo_scores;        /* [B,A,K] normalised */
    uint8_t* waymo_valid;       /* [B,S,A] */
    int32_t* mode_idx;          /* [B,A,K] selected input modes, or NULL */
    int64_t traj_strides[4];    /* element strides of `trajs` over (scene, agent, future, step), the D components adjacent; all 0 = the
                                 * contiguous [B,A,NP,S,D] layout.  Lets the caller hand over a rollout buffer where it lies -- preds
                                 * [B*K,A,S_all,4] as futures-of-a-scene, from the first future step on -- instead of a re-laid-out copy
                                 * (the reference's `trajs=buffer.preds[:, :, :, step_future_start:]`, waymo_motion.py:931-937) */
} tb_post_io;
int tb_post_process(tb_ctx* ctx, const tb_post_io* io, tb_stream stream);

/* Metric partials of a rollout buffer: the thirteen "sum" states of the reference's torchmetrics classes, in this order:
 *   ErrorMetrics (src/models/metrics/logging.py:9-54):        err_counter, err_pos_meter, err_rot_deg, err_spd_m_per_s
 *   TrafficRuleMetrics (:68-129): counter_agent, counter_veh, outside_map, collided, run_road_edge, run_red_light, passive,
 *                                 goal_reached, dest_reached
 * These are what a multi-GPU run SUM-all-reduces (dist_reduce_fx="sum"); the ratios of `compute()` are formed afterwards.
 * gt_valid / gt_states may be NULL (no ground truth: the four error sums are 0).  All masks uint8. */
typedef struct tb_metric_io {
    int32_t n_scene, n_agent, k_futures, n_step;
    int32_t loss_for_teacher_forcing;
    const uint8_t* pred_valid;      /* [B,A,K,S] */
    const float* pred_states;       /* [B,A,K,S,4] */
    const uint8_t* override_masks;  /* [B,A,K,S] */
    const uint8_t* gt_valid;        /* [B,A,S] or NULL */
    const float* gt_states;         /* [B,A,S,4] or NULL */
    const uint8_t* agent_role;      /* [B,A,3] */
    const int32_t* agent_type;      /* [B,A] */
    const uint8_t* outside_map;     /* [B,A,K,S] each */
    const uint8_t* collided;
    const uint8_t* run_road_edge;
    const uint8_t* run_red_light;
    const uint8_t* passive;
    const uint8_t* goal_reached;
    const uint8_t* dest_reached;
    double* out;                    /* [13] device buffer */
} tb_metric_io;
int tb_metric_partials(tb_ctx* ctx, const tb_metric_io* io, tb_stream stream);

/* -- the two samplers of joint_future_pred (waymo_motion.py:478-572) -------------------------------------------------------- */
/* Replaces: DiagGaussian / MyDist.sample + log_prob (src/models/modules/distributions.py:18-59) on the personality distribution that
 * LatentEncoder.forward returns (latent_encoder.py:70-147).  N = n_scene * k_futures, instance n uses the mean of scene n / k_futures
 * (what `latent.repeat_interleave_(k_futures, 0)` materialises, waymo_motion.py:493). */
typedef struct tb_latent_sample_io {
    int32_t n_scene, k_futures, n_agent;
    int32_t posterior;             /* nonzero: latent_post_dist.log_std, else latent_prior_dist.log_std (when log_std is NULL) */
    const float* log_std;          /* [16] device array, or NULL = the loaded parameter selected by `posterior` */
    const float* mean;             /* [B,A,16] */
    const float* eps;              /* [N,A,16] standard-normal draws (Normal.rsample's), or NULL = every agent takes the mean */
    const uint8_t* deterministic;  /* [N,A] or NULL: where set the agent takes the mean (the tensor form of `deterministic`) */
    const float* forced;           /* [N,A,16] or NULL: score THIS sample instead of drawing one (log_prob(sample)) */
    float* sample;                 /* [N,A,16] out, or NULL */
    float* log_prob;               /* [N,A] out, or NULL: Independent(Normal).log_prob of the sample */
} tb_latent_sample_io;
int tb_latent_sample(tb_ctx* ctx, const tb_latent_sample_io* io, tb_stream stream);

/* Replaces: DestCategorical.sample + log_prob (distributions.py:158-201) on the destination distribution of DestPredictor.forward
 * (goal_manager.py:202-333): arg max of the probabilities where deterministic, a draw elsewhere, and the log-prob of the result.
 * The draw is the inverse CDF of an EXPLICIT uniform number per (instance, agent) -- the reference calls torch.multinomial on torch's
 * global stream, which no other implementation can replay (its goldens force `goal_sample` instead: `forced`). */
typedef struct tb_dest_sample_io {
    int32_t n_scene, k_futures, n_agent, n_pl;
    /* 0: log_prob = logits - logsumexp(logits)  (Categorical(logits=), distributions.py:166-168)
     * 1: the distribution after `repeat_interleave_` (:196-199), Categorical(probs=softmax): log(clamp(probs, eps, 1 - eps)) */
    int32_t from_probs;
    const float* dest_logits;      /* [B,A,P] masked, un-normalised (tb_encode_io.dest_logits) */
    const float* uniform;          /* [N,A] draws in [0,1), or NULL = every agent takes the arg max */
    const uint8_t* deterministic;  /* [N,A] or NULL */
    const int32_t* forced;         /* [N,A] or NULL: score THESE destinations instead of choosing (log_prob(sample)) */
    int32_t* sample;               /* [N,A] out, or NULL */
    float* log_prob;               /* [N,A] out, or NULL */
    float* probs;                  /* [B,A,P] out, or NULL: softmax(dest_logits) (DestCategorical.probs) */
} tb_dest_sample_io;
int tb_dest_sample(tb_ctx* ctx, const tb_dest_sample_io* io, tb_stream stream);

/* sizeof() of the structs of this header as the library was compiled: out[0..5] = tb_config, tb_rollout_io, tb_encode_io,
 * tb_rule_io, tb_post_io, tb_metric_io; out[6] = the pointer size; out[7..8] = tb_posterior_io, tb_train_io; out[9..10] = tb_step_override, tb_forward_io;
 * out[11..12] = tb_latent_sample_io, tb_dest_sample_io -- lets a binding check its mirror of the layouts before the first call (no GPU needed). */
void tb_struct_sizes(int32_t out[13]);

/* Host-side helper (no device work, no context): bool one-hot rows [n_rows][n_class] (host memory) -> int32 class index per row, the
 * first set class, -1 where none is set -- the conversion SceneCentricPreProcessing does with `.argmax(-1)` on the device
 * (src/data_modules/scene_centric.py:92-133), done while a host batch is staged into pinned memory (trafficbots_amd/staging.py). */
void tb_host_onehot_index(const uint8_t* onehot, int64_t n_rows, int32_t n_class, int32_t* out);

#ifdef __cplusplus
}
#endif
#endif /* TRAFFICBOTS_HIP_H */
