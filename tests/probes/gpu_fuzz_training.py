"""Development probe (GPU box): `WaymoMotion.training_step` (forward value) with train-mode Bernoulli masks -- input / posterior history
dropout, hidden-state drop, explicit draws -- against the CPU oracle over random cases (the generator of tools/
fuzz_oracle_vs_reference.py --train, which holds the oracle to the reference's own training_step body).  Masks equal, personality means
2e-5, trajectories 1e-4 m (<= 40 steps), TrainingMetrics states rtol 2e-4.  usage: FUZZ_SEED=.. python tests/probes/gpu_fuzz_training.py 40"""
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

torch.set_num_threads(min(8, torch.get_num_threads()))
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(os.environ.get("FUZZ_SEED", "991")))
fails, worst = 0, {"mean": 0.0, "xy": 0.0}
for ci in range(n_cases):
    a, p_, t = int(rng.integers(2, 20)), int(rng.integers(3, 40)), int(rng.integers(1, 12))
    n_scene, step_end = int(rng.integers(1, 4)), int(rng.integers(15, 41))
    ov = {"pre_processing.input.dropout_p_history": float(rng.uniform(0.05, 0.5)), "pre_processing.latent.dropout_p_history": float(rng.uniform(0.05, 0.5)),
          "p_drop_hidden": float(rng.uniform(0.03, 0.3))}
    sc = dict(n_agent=a, n_pl=p_, n_tl=t, p_invalid_agent=float(rng.choice([0.0, 0.3])), p_late_spawn=float(rng.choice([0.0, 0.3])),
              p_future_spawn=float(rng.choice([0.0, 0.5])), p_future_exit=float(rng.choice([0.0, 0.3])), pos_range=float(rng.choice([40.0, 100.0])))
    seed, wseed, dseed = int(rng.integers(1, 2**30)), int(rng.integers(1, 1000)), int(rng.integers(1, 2**30))
    over = {"time_step_end": step_end, "n_joint_future": 1, **ov}
    cfg = load_model_config(overrides=over)
    n_step = step_end - cfg["time_step_sim_start"] + 1
    draws = synth.make_train_draws(dseed, n_scene, a, p_, t, n_step, ov["pre_processing.input.dropout_p_history"],
                                   ov["pre_processing.latent.dropout_p_history"], ov["p_drop_hidden"])
    batch = synth.make_val_batch(seed, n_scene, **sc)
    eps = synth.make_latent_noise(seed + 99, n_scene, a)
    sd = synth.make_state_dict(wseed)
    wm = WaymoMotion(**over)
    wm.load_state_dict(sd)
    keep = {k: torch.from_numpy(v) for k, v in draws.items() if k != "hidden_drop"}
    out = wm.training_step(batch, latent_eps=torch.from_numpy(eps).cuda(), history_keep=keep, hidden_drop=draws["hidden_drop"])
    torch.cuda.synchronize()
    buf = out["rollout_buffer"]
    with torch.no_grad():
        r = Oracle(sd, cfg, torch.float32).reactive_replay(batch, step_end, tf_cfg_name="teacher_forcing_training", eps=eps, history_keep=draws,
                                                           hidden_drop=draws["hidden_drop"])
    msgs = []
    for key, got, ref in (("post_valid", out["latent_post"].valid, r["post_valid"]), ("prior_valid", out["latent_prior"].valid, r["prior_valid"]),
                          ("valid", buf.valid, r["valid"]), ("override", buf.override_masks, r["override_masks"])):
        if not (got.cpu().bool().reshape(ref.shape) == ref.bool()).all():
            msgs.append(f"{key} differs")
    e_mean = max(float((out["latent_post"].mean.cpu() - r["post_mean"]).abs().max()), float((out["latent_prior"].mean.cpu() - r["prior_mean"]).abs().max()))
    v = r["valid"].bool()
    e_xy = float(((buf.preds.cpu().reshape(r["preds"].shape) - r["preds"]).abs() * v.unsqueeze(-1))[..., :2].max())
    if e_mean > 2e-5:
        msgs.append("personality tolerance")
    if e_xy > 1e-4:
        msgs.append("trajectory tolerance")
    # TrainingMetrics states from the ORACLE's buffer (the restated losses) against the HIP states
    s0 = cfg["time_step_sim_start"]
    gtv, gts = r["gt_valid"][:, s0: step_end + 1].transpose(1, 2), r["gt_state"][:, s0: step_end + 1].transpose(1, 2)
    rew, rv = TO.differentiable_reward(r["valid"], r["preds"], gtv, gts, r["agent_size"], cfg["differentiable_reward"])
    st = TO.training_metric_states(r["valid"], rv, rew, r["override_masks"], r["agent_role"], r["dest_logits_raw"], r["goal_valid"], r["gt_dest"],
                                   r["post_mean"], r["post_log_std"], r["post_valid"], r["prior_mean"], r["prior_log_std"], r["prior_valid"],
                                   cfg["training_metrics"])
    want = np.array([st[k] for k in TRAIN_FIELDS])
    got = out["train_states"].cpu().numpy()
    if not np.allclose(got, want, rtol=3e-4, atol=1e-5):
        msgs.append(f"train states {got} vs {want}")
    worst["mean"], worst["xy"] = max(worst["mean"], e_mean), max(worst["xy"], e_xy)
    fails += int(bool(msgs))
    print(f"case {ci:3d} B={n_scene} A={a:2d} P={p_:2d} T={t:2d} S={step_end} p=({ov['pre_processing.input.dropout_p_history']:.2f}, "
          f"{ov['pre_processing.latent.dropout_p_history']:.2f}, {ov['p_drop_hidden']:.2f}): means {e_mean:.1e} xy {e_xy:.1e}  {'ok' if not msgs else 'FAIL: ' + '; '.join(msgs)}", flush=True)
print(f"{n_cases} cases, {fails} failed; worst personality mean {worst['mean']:.2e}, trajectory {worst['xy']:.2e}")
sys.exit(1 if fails else 0)
