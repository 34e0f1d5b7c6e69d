"""Development probe (GPU box): the validation path (posterior, reactive replay from ground truth, rewards, TrainingMetrics states,
goal_reached) vs the CPU oracles over randomly drawn shapes, mask mixes and loss configurations.  One line per case; exits non-zero
on the first mismatch.  Not part of the test-suite."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import training_oracle as TO  # noqa: E402
from oracle.trafficbots_oracle import Oracle  # noqa: E402
from trafficbots_amd import synth  # noqa: E402
from trafficbots_amd.config import load_model_config  # noqa: E402
from trafficbots_amd.runtime import TRAIN_FIELDS  # noqa: E402
from trafficbots_amd.waymo_motion import WaymoMotion  # noqa: E402

torch.set_num_threads(min(8, torch.get_num_threads()))  # (the CPU oracles: small matrices, see tests/probes/oracle_thread_timing.py)
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 16
rng = np.random.default_rng(int(os.environ.get("FUZZ_SEED", "777")))
EDGE_A = [1, 2, 15, 16, 17, 31, 33, 48, 65]
EDGE_P = [2, 16, 31, 33, 64, 65, 130]
EDGE_T = [1, 2, 15, 33, 40]
worst = 0.0
n_outside = 0
for ci in range(n_cases):
    a, p, t = int(rng.choice(EDGE_A)), int(rng.choice(EDGE_P)), int(rng.choice(EDGE_T))
    n_scene, step_end = int(rng.integers(1, 4)), int(rng.choice([15, 30, 90]))
    scene = dict(n_agent=a, n_pl=p, n_tl=t, p_invalid_agent=float(rng.choice([0.0, 0.3, 0.7])), p_late_spawn=float(rng.choice([0.0, 0.3])),
                 p_early_exit=float(rng.choice([0.0, 0.3])), p_invalid_pl=float(rng.choice([0.0, 0.4])),
                 p_invalid_node=float(rng.choice([0.0, 0.5])), p_tl_valid=float(rng.choice([0.0, 0.3, 1.0])),
                 pos_range=float(rng.choice([30.0, 100.0, 148.0])), p_future_spawn=float(rng.choice([0.0, 0.5, 1.0])),
                 p_future_exit=float(rng.choice([0.0, 0.4])))
    over = {"time_step_end": step_end, "n_joint_future": int(rng.integers(1, 3)),
            "differentiable_reward.w_collision": float(rng.choice([0.0, 0.5])),
            "differentiable_reward.reduce_collsion_with_max": bool(rng.integers(0, 2)),
            "differentiable_reward.l_pos.criterion": str(rng.choice(["SmoothL1Loss", "MSELoss", "L1Loss"])),
            "differentiable_reward.l_rot.angular_type": [None, "cast", "cosine", "vector"][int(rng.integers(0, 4))],
            "training_metrics.loss_for_teacher_forcing": bool(rng.integers(0, 2)),
            "training_metrics.kl_for_unseen_agent": bool(rng.integers(0, 2)),
            "training_metrics.kl_balance_scale": float(rng.choice([-1.0, 0.8])),
            "training_metrics.kl_free_nats": float(rng.choice([-1.0, 0.01, 5.0])),
            "training_metrics.step_training_start": int(rng.choice([0, 10]))}
    if os.environ.get("FUZZ_CUR"):  # (round 6: a shorter history for the encoders -- FUZZ_CUR=5 -- on top of the draws above, which stay the same)
        over["time_step_current"] = int(os.environ["FUZZ_CUR"])
    seed = 30000 + ci
    if os.environ.get("FUZZ_ONLY") and ci != int(os.environ["FUZZ_ONLY"]):
        continue  # (after the draws, so that case ci is the same case as in a full run)
    cfg = load_model_config(overrides=over)
    sd = synth.make_state_dict(seed)
    batch = synth.make_val_batch(seed, n_scene, **scene)
    wm = WaymoMotion(**over)
    wm.load_state_dict(sd)
    out = wm.validation_step(batch)
    torch.cuda.synchronize()
    with torch.no_grad():
        r = Oracle(sd, cfg, torch.float32).reactive_replay(batch, step_end)
    buf = out["reactive_replay"]["rollout_buffer"]
    msgs = []
    e_post = float((out["latent_post"].mean.cpu() - r["post_mean"]).abs().max())
    if not (out["latent_post"].valid.cpu() == r["post_valid"]).all():
        msgs.append("post_valid differs")
    for key, got, ref in (("valid", buf.valid, r["valid"]), ("override", buf.override_masks, r["override_masks"]),
                          ("outside_map", buf.violations["outside_map"], r["outside_map"]),
                          ("dest_reached", buf.violations["dest_reached"], r["dest_reached"]),
                          ("goal_reached", buf.violations["goal_reached"], r["goal_reached"])):
        if not (got[:, :, 0].cpu() == ref).all():
            msgs.append(f"{key} differs")
    e_xy = float(((buf.preds[:, :, 0].cpu() - r["preds"]).abs() * r["valid"].unsqueeze(-1))[..., :2].max())
    gv = r["gt_valid"][:, 1: step_end + 1].transpose(1, 2)
    gs = r["gt_state"][:, 1: step_end + 1].transpose(1, 2)
    pv, ps = buf.valid[:, :, 0].cpu(), buf.preds[:, :, 0].cpu()
    rew, rv = TO.differentiable_reward(pv, ps, gv, gs, r["agent_size"], cfg["differentiable_reward"])
    if not (buf.diffbar_rewards_valid[:, :, 0].cpu() == rv).all():
        msgs.append("reward_valid differs")
    e_rew = float((buf.diffbar_rewards[:, :, 0].cpu() - rew).abs().max() / max(1.0, float(rew.abs().max())))
    st = TO.training_metric_states(pv, rv, rew, buf.override_masks[:, :, 0].cpu(), r["agent_role"], out["dest_logits"].cpu(), r["goal_valid"],
                                   r["gt_dest"], out["latent_post"].mean.cpu(), r["post_log_std"], out["latent_post"].valid.cpu(),
                                   out["latent_prior_mean"].cpu(), r["prior_log_std"], out["latent_prior_valid"].cpu(), cfg["training_metrics"])
    want = np.array([st.get(k, 0.0) for k in TRAIN_FIELDS])
    got = out["reactive_replay"]["train_states"].cpu().numpy()
    if not np.allclose(got, want, rtol=3e-5, atol=1e-6):
        msgs.append(f"train states {got} vs {want}")
    if not np.isfinite(buf.preds.cpu().numpy()).all():
        msgs.append("non-finite preds")
    if e_xy > 1e-4:
        # dense or long cases: the parity tests' ONE closed-loop rule (tools/ensemble.py::closed_loop_rule, frozen in round 4) against an
        # ensemble measured on the spot: 16 fp32 oracle runs on re-ordered batches with re-ordered Linear sums (Oracle(gemm_order_seed))
        from tools import ensemble

        with torch.no_grad():
            r64 = Oracle(sd, cfg, torch.float64).reactive_replay(batch, step_end)
            mem32, mem64 = [], []
            for mi in range(16):
                pb, perm = ensemble.permute_batch({k_: np.asarray(v) for k_, v in batch.items()}, 7919 * (ci + 1) + mi)
                rm = Oracle(sd, cfg, torch.float32, gemm_order_seed=4001 * (ci + 1) + mi).reactive_replay(pb, step_end)
                mp, mv = perm.agents_back(rm["preds"].numpy()), perm.agents_back(rm["valid"].numpy())
                mem32.append(ensemble.spread_per_step(mp, r["preds"].numpy(), mv & r["valid"].numpy(), 2))
                mem64.append(ensemble.spread_per_step(mp, r64["preds"].numpy(), mv & r64["valid"].numpy(), 2))
        v32, v64 = r["valid"].numpy(), r["valid"].numpy() & r64["valid"].numpy()
        hp = buf.preds[:, :, 0].cpu().numpy()
        d32 = ensemble.spread_per_step(hp, r["preds"].numpy(), v32, 2)
        d64 = ensemble.spread_per_step(hp, r64["preds"].numpy(), v64, 2)
        base64 = ensemble.spread_per_step(r["preds"].numpy(), r64["preds"].numpy(), v64, 2)
        rr_ = ensemble.closed_loop_rule(d32, d64, np.stack(mem32), np.stack([base64] + mem64))
        ok = rr_["ok"]
        print(f"        ensemble: oracle fp32 vs fp64 {base64.max():.1e} (members up to {rr_['members_max_vs_fp64']:.1e}); hip vs fp64 {d64.max():.1e} "
              f"(bound {rr_['bound_vs_fp64']:.1e}, rank {rr_['rank_vs_fp64']}), hip vs fp32 {d32.max():.1e} (bound {rr_['bound_vs_fp32']:.1e}, rank "
              f"{rr_['rank_vs_fp32']}) -> {'inside' if ok else 'OUTSIDE'}")
        if not ok:
            msgs.append("closed-loop envelope")
    if e_post > 2e-5 or e_rew > 1e-5:
        msgs.append("tolerance")
    worst = max(worst, e_xy)
    print(f"case {ci:2d} A={a:3d} P={p:3d} T={t:2d} B={n_scene} S={step_end} K={over['n_joint_future']}  post {e_post:.1e} xy {e_xy:.1e} reward {e_rew:.1e}"
          f"  {'ok' if not msgs else 'FAIL: ' + '; '.join(msgs)}", flush=True)
    if msgs:
        print(scene, over)
        # FUZZ_KEEP_GOING=1: a case that is ONLY outside the closed-loop envelope (a statistical statement about a chaotic loop, see
        # tests/probes/gpu_fuzz_case_detail.py) is counted and the run goes on; anything else still stops it
        if os.environ.get("FUZZ_KEEP_GOING") and msgs == ["closed-loop envelope"]:
            n_outside += 1
            continue
        sys.exit(1)
print(f"all {n_cases} cases ok; worst replay xy error {worst:.2e}" + (f"; {n_outside} outside the closed-loop envelope" if n_outside else ""))
