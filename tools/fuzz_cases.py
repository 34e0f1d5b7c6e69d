"""Random case generator shared by tools/fuzz_oracle_vs_reference.py (oracle vs the imported reference, here) and
tests/probes/gpu_fuzz_cases.py (HIP vs the oracle, GPU box): shapes, mask densities, K, horizon, weight distributions, scenes at the edge
of the layout (synth.EDGE_SETS), scalar config overrides (time_step_current 5 / 10, teacher-forcing steps, action bounds per class),
sampled actions, rule checks.  No import of the reference here."""


def draw_case(rng) -> dict:
    a, p, t = int(rng.integers(1, 18)), int(rng.integers(2, 40)), int(rng.integers(1, 12))
    scene = dict(n_agent=a, n_pl=p, n_tl=t, p_invalid_agent=float(rng.choice([0.0, 0.2, 0.6])), p_late_spawn=float(rng.choice([0.0, 0.3])),
                 p_early_exit=float(rng.choice([0.0, 0.2])), p_invalid_pl=float(rng.choice([0.0, 0.3])),
                 p_invalid_node=float(rng.choice([0.0, 0.5])), p_tl_valid=float(rng.choice([0.0, 0.3, 0.8])),
                 pos_range=float(rng.choice([25.0, 100.0, 400.0])), spd_max=float(rng.choice([1.0, 15.0, 30.0])))
    edge = rng.choice(["", "", "v1", "v2", "v3"])
    if edge:
        scene["edge"] = str(edge)
    over = {}
    if rng.random() < 0.4:
        cur = int(rng.choice([5, 10]))
        over["time_step_current"] = cur
        over["teacher_forcing_joint_future_pred.step_warm_start"] = int(rng.integers(0, cur + 1))
        over["teacher_forcing_joint_future_pred.step_spawn_agent"] = int(rng.integers(0, 11))  # (may exceed `cur`: the history has 11 steps)
    if rng.random() < 0.4:
        for cls in ("veh", "cyc", "ped"):
            over[f"dynamics.{cls}.max_acc"] = float(rng.uniform(2.0, 8.0))
            over[f"dynamics.{cls}.max_yaw_rate"] = float(rng.uniform(0.5, 7.0))
    case = dict(base_seed=int(rng.integers(1, 2**30)), n_scene=int(rng.integers(1, 4)), k=int(rng.integers(1, 4)),
                weight_seed=int(rng.integers(1, 1000)), time_step_end=int(rng.integers(12, 46)), scene=scene, tap_steps=[], overrides=over,
                fp64=True, store_feats=True)
    if rng.random() < 0.3:
        case["weight_mode"] = str(rng.choice(["normal", "sharp", "ln_gamma"]))
    if rng.random() < 0.25:
        case["action_noise"] = True
    if rng.random() < 0.25:
        case["rule_flags"] = True
    if rng.random() < 0.2:  # `forward(action_override=, mask_action_override=)` at every step (dynamics.py:96-100; the stepwise API of the mirror)
        case["action_override"] = True
    return case


