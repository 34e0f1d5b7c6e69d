"""Development probe (GPU box): the HIP path against the CPU oracle over the random cases of tools/fuzz_cases.py -- the generator the
oracle itself is held to the imported reference with (tools/fuzz_oracle_vs_reference.py: 200 cases, fp64, 1e-9): shapes, masks, K,
horizon, weight distributions, scenes at the edge of the layout, config overrides (time_step_current 5 / 10, teacher-forcing steps,
action bounds per class), sampled actions.  Same destinations (the HIP run's draws), same personality noise; masks and flags EQUAL,
one-shot tensors 2e-5 (4e-6 of the largest feature entry where that is more), trajectories 1e-4 m over the (short) horizons, beyond that the
closed-loop rule of the parity tests against an oracle ensemble measured on the spot.  usage: FUZZ_SEED=.. python tests/probes/gpu_fuzz_cases.py 80"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from fuzz_cases import draw_case  # noqa: E402
from oracle.trafficbots_oracle import Oracle  # noqa: E402
from trafficbots_amd import synth  # noqa: E402
from trafficbots_amd.config import load_model_config  # noqa: E402
from trafficbots_amd.waymo_motion import WaymoMotion  # noqa: E402

torch.set_num_threads(min(8, torch.get_num_threads()))
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(os.environ.get("FUZZ_SEED", "777")))
worst = {"xy": 0.0, "one_shot": 0.0, "logp": 0.0}
fails = 0
for ci in range(n_cases):
    case = draw_case(rng)
    if os.environ.get("FUZZ_ONLY") and ci != int(os.environ["FUZZ_ONLY"]):
        continue  # (after the draw, so that case ci is the same case as in a full run)
    case.pop("rule_flags", None)  # (flag-gated checks: tests/probes/gpu_fuzz_rules_post_metrics.py)
    case.pop("action_override", None)  # (per-call action overrides belong to the stepwise API: tests/test_gpu_boundary.py)
    sc, k, n_scene, step_end = case["scene"], case["k"], case["n_scene"], case["time_step_end"]
    over = {"time_step_end": step_end, "n_joint_future": k, **case["overrides"]}
    cfg = load_model_config(overrides=over)
    sd = synth.case_state_dict(case)
    batch = synth.make_batch(case["base_seed"], n_scene, **sc)
    n, a = n_scene * k, sc["n_agent"]
    eps = synth.make_latent_noise(case["base_seed"] + 99, n, a)
    act = None
    if case.get("action_noise"):
        act = synth.make_action_noise(case["base_seed"] + 77, n, a, step_end - cfg["time_step_sim_start"] + 1)
    wm = WaymoMotion(**over, **({"operand_precision": os.environ["FUZZ_PRECISION"]} if os.environ.get("FUZZ_PRECISION") else {}))
    wm.load_state_dict(sd)
    out = wm.test_step(batch, latent_eps=torch.from_numpy(eps).cuda(), generator=torch.Generator(device="cuda").manual_seed(case["base_seed"] % 2**31),
                       action_eps=None if act is None else torch.from_numpy(act).cuda())
    torch.cuda.synchronize()
    buf = out["rollout_buffer"]
    dest = out["goal_sample"].transpose(1, 2).reshape(n, -1).cpu().numpy()
    with torch.no_grad():
        r = Oracle(sd, cfg, torch.float32, hoist=True).joint_future_pred(batch, k, eps, step_end, dest_override=dest, action_eps=act)
    msgs = []
    for key, got, ref in (("valid", buf.valid, r["valid"]), ("override", buf.override_masks, r["override_masks"]),
                          ("outside_map", buf.violations["outside_map"], r["outside_map"]),
                          ("dest_reached", buf.violations["dest_reached"], r["dest_reached"]),
                          ("latent_valid", out["latent_valid"], r["latent_valid"])):
        if not (got.cpu().bool() == ref.bool()).all():
            msgs.append(f"{key} differs ({int((got.cpu().bool() != ref.bool()).sum())})")
    f = out["input_feature_dict"]
    e_one = max(float((f["map_feature"].cpu() - r["map_feature"]).abs().max()), float((out["latent_mean"].cpu() - r["latent_mean"]).abs().max()),
                float((f["agent_feature"][:, -1].cpu() - r["agent_feature_cur"]).abs().max()),
                float((f["tl_feature"][:, -1].cpu() - r["tl_feature_cur"]).abs().max()))
    lg = torch.log_softmax(out["dest_logits"], -1).cpu()
    fin = torch.isfinite(r["dest_logits"])
    if not (torch.isfinite(lg) == fin).all():
        msgs.append("destination candidate mask differs")
    v = r["valid"].bool()
    e_logp = max(float((out["goal_log_probs"].cpu() - r["goal_log_probs"]).abs().max()),
                 float((buf.latent_log_probs.cpu() - r["latent_log_probs"]).abs().max()),
                 float(((buf.action_log_probs.cpu() - r["action_log_probs"]).abs() * v).max()))
    e_xy = float(((buf.preds.cpu() - r["preds"]).abs() * v.unsqueeze(-1))[..., :2].max())
    if not torch.isfinite(buf.preds).all():
        msgs.append("non-finite preds")
    # (2e-5 on O(1) values; a feature tensor whose entries reach tens -- `sharp` / `ln_gamma` weights -- is held to the same RELATIVE
    # accuracy, 4e-6 of its largest entry: the rule of tests/test_gpu_parity.py)
    scale = max(float(r["map_feature"].abs().max()), float(r["agent_feature_cur"].abs().max()), float(r["tl_feature_cur"].abs().max()))
    if e_one > max(2e-5, 4e-6 * scale):
        msgs.append(f"one-shot tolerance (largest feature entry {scale:.3g})")
    if e_logp > 2e-4:
        msgs.append("log-prob tolerance")
    if os.environ.get("FUZZ_VERBOSE"):
        with torch.no_grad():
            r64v = Oracle(sd, cfg, torch.float64).joint_future_pred(batch, k, eps, step_end, dest_override=dest, action_eps=act)
        d32 = ((buf.preds.cpu() - r["preds"]).abs() * v.unsqueeze(-1))[..., :2].amax(dim=(0, 1, 2, 4))
        d64 = ((buf.preds.cpu().double() - r64v["preds"]).abs() * v.unsqueeze(-1))[..., :2].amax(dim=(0, 1, 2, 4))
        o64 = ((r["preds"].double() - r64v["preds"]).abs() * v.unsqueeze(-1))[..., :2].amax(dim=(0, 1, 2, 4))
        print("   per step |hip - oracle32|:", " ".join(f"{x:.0e}" for x in d32.tolist()))
        print("   per step |hip - oracle64|:", " ".join(f"{x:.0e}" for x in d64.tolist()))
        print("   per step |o32 - oracle64|:", " ".join(f"{x:.0e}" for x in o64.tolist()))
    if e_xy > 1e-4:
        # beyond north_star's flat bound: the parity tests' ONE closed-loop rule (tools/ensemble.py::closed_loop_rule) against an ensemble
        # measured on the spot -- 16 fp32 oracle runs on re-ordered batches with re-ordered Linear sums -- as tests/probes/
        # gpu_fuzz_validation.py does
        from tools import ensemble

        with torch.no_grad():
            r64 = Oracle(sd, cfg, torch.float64).joint_future_pred(batch, k, eps, step_end, dest_override=dest, action_eps=act)
            mem32, mem64 = [], []
            for mi in range(16):
                pb, perm = ensemble.permute_batch({k_: np.asarray(v_) for k_, v_ in batch.items()}, 7919 * (ci + 1) + mi)
                rm = Oracle(sd, cfg, torch.float32, gemm_order_seed=4001 * (ci + 1) + mi).joint_future_pred(
                    pb, k, perm.agents_fwd(eps, k), step_end, dest_override=perm.dest_fwd(dest, k),
                    action_eps=None if act is None else perm.agents_fwd(act, k))
                mp, mv = perm.agents_back(rm["preds"].numpy(), 1), perm.agents_back(rm["valid"].numpy(), 1)
                mem32.append(ensemble.spread_per_step(mp, r["preds"].numpy(), mv & r["valid"].numpy(), 3))
                mem64.append(ensemble.spread_per_step(mp, r64["preds"].numpy(), mv & r64["valid"].numpy(), 3))
        hp = buf.preds.cpu().numpy()
        v64 = r["valid"].numpy() & r64["valid"].numpy()
        d32 = ensemble.spread_per_step(hp, r["preds"].numpy(), r["valid"].numpy(), 3)
        d64 = ensemble.spread_per_step(hp, r64["preds"].numpy(), v64, 3)
        base64 = ensemble.spread_per_step(r["preds"].numpy(), r64["preds"].numpy(), v64, 3)
        rr_ = ensemble.closed_loop_rule(d32, d64, np.stack(mem32), np.stack([base64] + mem64))
        print(f"        ensemble: oracle fp32 vs fp64 {base64.max():.1e} (members up to {rr_['members_max_vs_fp64']:.1e}); hip vs fp64 {d64.max():.1e} "
              f"(bound {rr_['bound_vs_fp64']:.1e}, rank {rr_['rank_vs_fp64']}), hip vs fp32 {d32.max():.1e} (bound {rr_['bound_vs_fp32']:.1e}, rank "
              f"{rr_['rank_vs_fp32']}) -> {'inside' if rr_['ok'] else 'OUTSIDE'}")
        if not rr_["ok"]:
            msgs.append("closed-loop envelope")
    for kk, vv in (("xy", e_xy), ("one_shot", e_one), ("logp", e_logp)):
        worst[kk] = max(worst[kk], vv)
    fails += int(bool(msgs))
    print(f"case {ci:3d} B={n_scene} K={k} A={a:2d} P={sc['n_pl']:2d} T={sc['n_tl']:2d} S={step_end} edge={sc.get('edge', '-')} w={case.get('weight_mode', 'default')} "
          f"{'act-noise ' if act is not None else ''}over={sorted(x.split('.')[-1] for x in case['overrides'])}: one-shot {e_one:.1e} logp {e_logp:.1e} xy {e_xy:.1e}  "
          f"{'ok' if not msgs else 'FAIL: ' + ', '.join(msgs)}", flush=True)
print(f"{n_cases} cases, {fails} failed; worst one-shot {worst['one_shot']:.2e}, log-prob {worst['logp']:.2e}, closed-loop xy {worst['xy']:.2e}")
sys.exit(1 if fails else 0)
