#!/usr/bin/env python3
"""Differential fuzz of the CPU oracle against THE REFERENCE itself, both in fp64 (runs here only: the reference does not travel).

The committed goldens pin the oracle at 36 hand-picked cases; this draws random ones -- shapes, mask densities, K, horizon, weight
distributions, scenes at the edge of the layout (synth.EDGE_SETS), scalar config overrides, sampled actions, rule checks -- runs the
imported reference (`tools/gen_golden.py::run_reference`, fp64) and the oracle (fp64) on the same inputs, the same personality noise and
the reference's own destination draws, and compares EVERYTHING the golden tests compare.  In fp64 both are the same function up to
summation order, so the bound is tight (1e-9 on trajectories over the whole closed loop, masks and flags equal): a semantic deviation
of the oracle cannot hide behind rounding noise or chaos.  Usage: python tools/fuzz_oracle_vs_reference.py [n_cases] [seed]
-> profiles/r06_oracle_vs_reference_fuzz.txt"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

import gen_golden  # noqa: E402
from oracle.trafficbots_oracle import Oracle  # noqa: E402
from trafficbots_amd import synth  # noqa: E402
from trafficbots_amd.config import load_model_config  # noqa: E402

BOOL_KEYS = ("valid", "override_masks", "dest_reached", "outside_map", "outside_map_this_step", "dest_reached_this_step",
             "latent_valid", "map_feature_valid", "final_valid")
ONE_SHOT = ("latent_mean", "latent_sample", "map_feature", "agent_feature_cur", "agent_feature_0", "tl_feature_cur", "goal_log_probs",
            "latent_log_probs", "action_log_probs")


from fuzz_cases import draw_case  # noqa: E402,F401


def compare(case: dict) -> dict:
    ref = gen_golden.run_reference(case, torch.float64)
    over = {"time_step_end": case["time_step_end"], "n_joint_future": case["k"], **case.get("overrides", {})}
    if case.get("rule_flags"):
        over["traffic_rule_checker"] = {f"enable_check_{c}": True for c in ("collided", "run_road_edge", "run_red_light", "passive")}
    cfg = load_model_config(overrides=over)
    sd = synth.case_state_dict(case)
    sc = case["scene"]
    batch = synth.make_batch(case["base_seed"], case["n_scene"], **sc)
    n = case["n_scene"] * case["k"]
    eps = synth.make_latent_noise(case["base_seed"] + 99, n, sc["n_agent"])
    dest = np.transpose(ref["goal_sample"], (0, 2, 1)).reshape(n, -1)
    act = None
    if case.get("action_noise"):
        act = synth.make_action_noise(case["base_seed"] + 77, n, sc["n_agent"], case["time_step_end"] - cfg["time_step_sim_start"] + 1)
    act_ov = None
    if case.get("action_override"):
        act_ov = synth.make_action_override(case["base_seed"] + 55, n, sc["n_agent"], case["time_step_end"] - cfg["time_step_sim_start"] + 1)
    with torch.no_grad():
        r = Oracle(sd, cfg, torch.float64).joint_future_pred(batch, case["k"], eps, case["time_step_end"], dest_override=dest, action_eps=act,
                                                             action_override=act_ov)
    rep = {}
    for k in BOOL_KEYS:
        if k in ref and k in r:
            rep[k] = int((r[k].numpy() != ref[k]).sum())
    v = ref["valid"][..., None]
    rep["preds"] = float((np.abs(r["preds"].numpy() - ref["preds"]) * v).max())
    for k in ONE_SHOT:
        if k in ref and k in r:
            x, y = r[k].numpy(), ref[k]
            fin = np.isfinite(y)
            rep[k] = float(np.abs(np.where(fin, x - y, 0)).max()) if fin.any() else 0.0
            rep[k + "/finite_mask"] = int((np.isfinite(x) != fin).sum())
    fin = np.isfinite(ref["dest_logits"])
    rep["dest_logits/finite_mask"] = int((np.isfinite(r["dest_logits"].numpy()) != fin).sum())
    rep["dest_logits"] = float(np.abs(np.where(fin, r["dest_logits"].numpy() - ref["dest_logits"], 0)).max())
    if case.get("rule_flags"):  # the four flag-gated checks on the (valid, state) pairs the reference handed to its checker
        from oracle.rule_checks_oracle import rule_checks

        b = {k: torch.from_numpy(np.asarray(v)) for k, v in batch.items()}
        res = rule_checks(torch.from_numpy(ref["check_state"]), torch.from_numpy(ref["check_valid"]), case["k"], cfg["time_step_sim_start"],
                          b["history/agent/type"], b["history/agent/size"], b["map/valid"], b["map/type"], b["map/pos"], b["map/dir"],
                          b["history/tl_stop/valid"][:, : cfg["time_step_current"] + 1], b["history/tl_stop/pos"][:, : cfg["time_step_current"] + 1],
                          b["history/tl_stop/state"][:, : cfg["time_step_current"] + 1])
        for k, v in res.items():
            want = np.transpose(ref[k], (0, 2, 1, 3)).reshape(v.shape)  # [B,A,K,S] -> [N,A,S]
            rep["rule/" + k] = int((v.numpy() != want).sum())
            rep["rule/" + k + "/raised"] = -int(want.sum())  # (negative: informational, not a mismatch count)
    return rep


TRAIN_FIELDS = ("vae_kl_counter", "vae_kl", "diffbar_reward_counter", "diffbar_reward", "goal_loss", "goal_counter")


def draw_val_case(rng) -> dict:
    c = draw_case(rng)
    sc = c["scene"]
    sc.update(p_future_spawn=float(rng.choice([0.0, 0.4, 0.8])), p_future_exit=float(rng.choice([0.0, 0.3])))
    sc["n_pl"] = max(sc["n_pl"], 3)
    over = {k: v for k, v in c["overrides"].items() if k.startswith("dynamics.") or k == "time_step_current"}  # (the replay's warm start follows it)
    if rng.random() < 0.5:
        over.update({"differentiable_reward.w_collision": float(rng.choice([0.0, 0.5, 1.0])),
                     "differentiable_reward.reduce_collsion_with_max": bool(rng.random() < 0.5),
                     "differentiable_reward.l_pos.criterion": str(rng.choice(["SmoothL1Loss", "MSELoss", "L1Loss"])),
                     "differentiable_reward.l_rot.angular_type": str(rng.choice(["cosine", "cast", "vector"])),
                     "training_metrics.loss_for_teacher_forcing": bool(rng.random() < 0.5),
                     "training_metrics.kl_for_unseen_agent": bool(rng.random() < 0.5),
                     "training_metrics.kl_balance_scale": float(rng.choice([-1.0, 0.8])),
                     "training_metrics.kl_free_nats": float(rng.choice([-1.0, 0.01]))})
    extra = {}
    if rng.random() < 0.3:  # training.py:85-89: agents without a role keep their loss terms only where a Bernoulli draw says so (stored draw)
        over["training_metrics.p_loss_for_irrelevant"] = float(rng.choice([0.3, 0.6]))
        extra["irrelevant_seed"] = int(rng.integers(1, 2**30))
    return dict(base_seed=c["base_seed"], n_scene=c["n_scene"], weight_seed=c["weight_seed"], time_step_end=int(rng.integers(15, 91)),
                overrides=over, scene=sc, fp64=True, **extra, **({"weight_mode": c["weight_mode"]} if "weight_mode" in c else {}))


def compare_val(case: dict) -> dict:
    import gen_golden_val
    from oracle import training_oracle as TO

    ref = gen_golden_val.run_reference(case, torch.float64, with_jfp=True)
    over = {"time_step_end": case["time_step_end"], "n_joint_future": 1, **case["overrides"]}
    cfg = load_model_config(overrides=over)
    sd = synth.case_state_dict(case)
    batch = synth.make_val_batch(case["base_seed"], case["n_scene"], **case["scene"])
    with torch.no_grad():
        r = Oracle(sd, cfg, dtype=torch.float64).reactive_replay(batch, case["time_step_end"])
    rep = {}
    for k in ("post_valid", "prior_valid", "valid", "override_masks", "dest_reached", "outside_map", "goal_reached"):
        rep[k] = int((r[k].numpy() != ref[k]).sum())
    for k in ("post_mean", "prior_mean", "latent_log_probs"):
        rep[k] = float(np.abs(r[k].numpy() - ref[k]).max())
    rep["preds"] = float((np.abs(r["preds"].numpy() - ref["preds"]) * ref["valid"][..., None]).max())
    s0, s1 = cfg["time_step_sim_start"], case["time_step_end"]
    gtv, gts = r["gt_valid"][:, s0: s1 + 1].transpose(1, 2), r["gt_state"][:, s0: s1 + 1].transpose(1, 2)
    rew, rv = TO.differentiable_reward(r["valid"], r["preds"], gtv, gts, r["agent_size"], cfg["differentiable_reward"])
    rep["diffbar_rewards_valid"] = int((rv.numpy() != ref["diffbar_rewards_valid"]).sum())
    both = rv.numpy() & ref["diffbar_rewards_valid"]
    rep["diffbar_rewards"] = float(np.abs(np.where(both, rew.numpy() - ref["diffbar_rewards"], 0)).max())
    st = TO.training_metric_states(r["valid"], rv, rew, r["override_masks"], r["agent_role"], r["dest_logits_raw"], r["goal_valid"], r["gt_dest"],
                                   r["post_mean"], r["post_log_std"], r["post_valid"], r["prior_mean"], r["prior_log_std"], r["prior_valid"],
                                   cfg["training_metrics"],
                                   irrelevant_draw=torch.from_numpy(ref["irrelevant_draw"]) if "irrelevant_draw" in ref else None)
    got = np.array([st[k] for k in TRAIN_FIELDS], np.float64)
    rep["train_states_rel"] = float(np.max(np.abs(got - ref["train_states"]) / np.maximum(1.0, np.abs(ref["train_states"]))))
    # second half of validation_step: joint_future_pred on the 91-step ground truth (`Oracle.joint_future_pred(use_gt=True)`)
    dest = np.transpose(ref["jfp/goal_sample"], (0, 2, 1)).reshape(case["n_scene"], -1)
    with torch.no_grad():
        j = Oracle(sd, cfg, dtype=torch.float64).joint_future_pred(batch, 1, None, case["time_step_end"], dest_override=dest, use_gt=True)
    own = j["dest_logits_raw"].argmax(-1).numpy()
    rep["jfp/argmax_dest"] = int((own != ref["jfp/goal_sample"][:, :, 0]).sum())
    for k in ("valid", "override_masks", "outside_map", "dest_reached", "goal_reached"):
        rep["jfp/" + k] = int((j[k].numpy() != ref["jfp/" + k]).sum())
    jv = ref["jfp/valid"][..., None]
    rep["jfp/preds"] = float((np.abs(j["preds"].numpy() - ref["jfp/preds"]) * jv).max())
    rep["jfp/action_log_probs"] = float((np.abs(j["action_log_probs"].numpy() - ref["jfp/action_log_probs"]) * ref["jfp/valid"]).max())
    rep["jfp/latent_log_probs"] = float(np.abs(j["latent_log_probs"].numpy() - ref["jfp/latent_log_probs"]).max())
    return rep


def main_train() -> int:
    """Train-mode Bernoulli masks (input / posterior history dropout, hidden-state drop) with explicit draws: the body of the reference's
    training_step (tools/gen_golden_val.py::run_reference_training, fp32) against `Oracle.reactive_replay(history_keep=, hidden_drop=)`."""
    import gen_golden_val

    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 20260932)
    lines, bad, t0 = [], 0, time.time()
    for i in range(n_cases):
        a, p_, t = int(rng.integers(2, 14)), int(rng.integers(3, 30)), int(rng.integers(1, 12))
        case = dict(base_seed=int(rng.integers(1, 2**30)), n_scene=int(rng.integers(1, 4)), weight_seed=int(rng.integers(1, 1000)),
                    time_step_end=int(rng.integers(15, 41)), draws_seed=int(rng.integers(1, 2**30)),
                    overrides={"pre_processing.input.dropout_p_history": float(rng.uniform(0.05, 0.5)),
                               "pre_processing.latent.dropout_p_history": float(rng.uniform(0.05, 0.5)), "p_drop_hidden": float(rng.uniform(0.03, 0.3))},
                    scene=dict(n_agent=a, n_pl=p_, n_tl=t, p_invalid_agent=float(rng.choice([0.0, 0.3])), p_late_spawn=float(rng.choice([0.0, 0.3])),
                               p_future_spawn=float(rng.choice([0.0, 0.5])), p_future_exit=float(rng.choice([0.0, 0.3])), pos_range=float(rng.choice([40.0, 100.0]))))
        tag = f"train case {i:3d} B={case['n_scene']} A={a:2d} P={p_:2d} T={t:2d} S={case['time_step_end']} p=({case['overrides']['pre_processing.input.dropout_p_history']:.2f}, {case['overrides']['pre_processing.latent.dropout_p_history']:.2f}, {case['overrides']['p_drop_hidden']:.2f})"
        try:
            g = gen_golden_val.run_reference_training(case)
            over = {"time_step_end": case["time_step_end"], "n_joint_future": 1, **case["overrides"]}
            cfg = load_model_config(overrides=over)
            sc, ov = case["scene"], case["overrides"]
            n_step = case["time_step_end"] - cfg["time_step_sim_start"] + 1
            draws = synth.make_train_draws(case["draws_seed"], case["n_scene"], sc["n_agent"], sc["n_pl"], sc["n_tl"], n_step,
                                           ov["pre_processing.input.dropout_p_history"], ov["pre_processing.latent.dropout_p_history"], ov["p_drop_hidden"])
            batch = synth.make_val_batch(case["base_seed"], case["n_scene"], **sc)
            eps = synth.make_latent_noise(case["base_seed"] + 99, case["n_scene"], sc["n_agent"])
            with torch.no_grad():
                r = Oracle(synth.make_state_dict(case["weight_seed"]), cfg, torch.float32).reactive_replay(
                    batch, case["time_step_end"], tf_cfg_name="teacher_forcing_training", eps=eps, history_keep=draws, hidden_drop=draws["hidden_drop"])
        except Exception as e:
            lines.append(f"{tag}: NOT RUN ({type(e).__name__}: {str(e)[:120]})")
            print(lines[-1], flush=True)
            continue
        masks = sum(int((r[k].numpy() != g[k]).sum()) for k in ("post_valid", "prior_valid", "valid", "override_masks"))
        e_mean = max(float(np.abs(r["post_mean"].numpy() - g["post_mean"]).max()), float(np.abs(r["prior_mean"].numpy() - g["prior_mean"]).max()))
        e_xy = float((np.abs(r["preds"].numpy() - g["preds"]) * g["valid"][..., None]).max())
        ok = masks == 0 and e_mean <= 2e-6 and e_xy <= 1e-4
        bad += int(not ok)
        lines.append(f"{tag}: {'ok' if ok else 'DIFFERS'} masks {masks} means {e_mean:.1e} preds {e_xy:.1e}")
        print(lines[-1], flush=True)
    head = [f"# tools/fuzz_oracle_vs_reference.py --train: train-mode masks with explicit draws, the reference's training_step body (fp32) against the oracle (fp32);",
            f"# masks equal, personality means <= 2e-6, trajectories <= 1e-4 m (<= 40 steps).  {n_cases - sum('NOT RUN' in l for l in lines)} cases run, {bad} differ; {time.time() - t0:.0f} s."]
    open(os.path.join(ROOT, "profiles", "r06_oracle_vs_reference_fuzz_train.txt"), "w").write("\n".join(head + lines) + "\n")
    print("\n".join(head))
    return 1 if bad else 0


def main() -> int:
    if "--train" in sys.argv:
        sys.argv.remove("--train")
        return main_train()
    if "--val" in sys.argv:
        sys.argv.remove("--val")
        return main_val()
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 20260930
    rng = np.random.default_rng(seed)
    lines, bad = [], 0
    t0 = time.time()
    worst = {}
    for i in range(n_cases):
        case = draw_case(rng)
        tag = (f"case {i:3d} B={case['n_scene']} K={case['k']} A={case['scene']['n_agent']:2d} P={case['scene']['n_pl']:2d} T={case['scene']['n_tl']:2d} "
               f"S={case['time_step_end']} edge={case['scene'].get('edge', '-')} w={case.get('weight_mode', 'default')} "
               f"{'act-noise ' if case.get('action_noise') else ''}{'act-override ' if case.get('action_override') else ''}{'rules ' if case.get('rule_flags') else ''}over={sorted(k.split('.')[-1] for k in case['overrides'])}")
        try:
            rep = compare(case)
        except Exception as e:  # a case the reference itself cannot run is reported, not counted
            lines.append(f"{tag}: NOT RUN ({type(e).__name__}: {str(e)[:100]})")
            continue
        ints = {k: v for k, v in rep.items() if isinstance(v, int) and v > 0}
        flts = {k: v for k, v in rep.items() if isinstance(v, float)}
        for k, v in flts.items():
            worst[k] = max(worst.get(k, 0.0), v)
        ok = not ints and all(v <= 1e-9 for v in flts.values())
        bad += int(not ok)
        lines.append(f"{tag}: {'ok' if ok else 'DIFFERS'} preds {rep['preds']:.1e}" + ("" if ok else f"  {json.dumps({**ints, **{k: v for k, v in flts.items() if v > 1e-9}})}"))
        print(lines[-1], flush=True)
    head = [f"# tools/fuzz_oracle_vs_reference.py {n_cases} {seed}: the CPU oracle against the imported reference, both fp64, random cases (shapes, masks, K,",
            "# horizon, weight distributions, edge scenes, config overrides, sampled actions, rule checks); bound 1e-9 on every float tensor incl. the",
            f"# closed-loop trajectories, masks / flags / finite-masks equal.  {n_cases - sum('NOT RUN' in l for l in lines)} cases run, {bad} differ; {time.time() - t0:.0f} s.",
            "# worst absolute difference per tensor over all cases: " + ", ".join(f"{k} {v:.1e}" for k, v in sorted(worst.items()))]
    out = os.path.join(ROOT, "profiles", "r06_oracle_vs_reference_fuzz.txt")
    open(out, "w").write("\n".join(head + lines) + "\n")
    print("\n".join(head))
    return 1 if bad else 0




def main_val() -> int:
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 20260931
    rng = np.random.default_rng(seed)
    lines, bad, worst, t0 = [], 0, {}, time.time()
    for i in range(n_cases):
        case = draw_val_case(rng)
        sc = case["scene"]
        tag = (f"val case {i:3d} B={case['n_scene']} A={sc['n_agent']:2d} P={sc['n_pl']:2d} T={sc['n_tl']:2d} S={case['time_step_end']} edge={sc.get('edge', '-')} "
               f"w={case.get('weight_mode', 'default')} over={sorted(k.split('.')[-1] for k in case['overrides'])}")
        try:
            rep = compare_val(case)
        except Exception as e:
            lines.append(f"{tag}: NOT RUN ({type(e).__name__}: {str(e)[:120]})")
            print(lines[-1], flush=True)
            continue
        ints = {k: v for k, v in rep.items() if isinstance(v, int) and v > 0}
        flts = {k: v for k, v in rep.items() if isinstance(v, float)}
        for k, v in flts.items():
            worst[k] = max(worst.get(k, 0.0), v)
        # (the reference keeps its metric states in float32 accumulators whatever the model's dtype: 1e-6 relative there)
        ok = not ints and all(v <= (1e-6 if k == "train_states_rel" else 1e-9) for k, v in flts.items())
        bad += int(not ok)
        lines.append(f"{tag}: {'ok' if ok else 'DIFFERS'} preds {rep['preds']:.1e}" + ("" if ok else f"  {json.dumps({**ints, **{k: v for k, v in flts.items() if v > 1e-9}})}"))
        print(lines[-1], flush=True)
    head = [f"# tools/fuzz_oracle_vs_reference.py --val {n_cases} {seed}: the validation path (posterior / prior personalities, reactive replay from the ground truth,",
            "# differentiable rewards, TrainingMetrics states; then joint_future_pred on the 91-step ground truth -- kill rule, goal_reached -- as the second",
            "# half of validation_step runs it) of the CPU oracles against the imported reference, both fp64, random cases; bound 1e-9",
            "# (TrainingMetrics states: 1e-6 relative, the reference accumulates them in float32).",
            f"# {n_cases - sum('NOT RUN' in l for l in lines)} cases run, {bad} differ; {time.time() - t0:.0f} s.",
            "# worst difference per quantity: " + ", ".join(f"{k} {v:.1e}" for k, v in sorted(worst.items()))]
    out = os.path.join(ROOT, "profiles", "r06_oracle_vs_reference_fuzz_val.txt")
    open(out, "w").write("\n".join(head + lines) + "\n")
    print("\n".join(head))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
