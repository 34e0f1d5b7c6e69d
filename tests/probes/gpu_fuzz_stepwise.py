"""Development probe (GPU box): the STEPWISE interface -- `wm.rollout(..., stepwise=True)` + one `wm.forward(...)` per simulation step
(`tb_rollout_begin / tb_rollout_step_ex / tb_rollout_state`), with per-call action overrides on random (agent, step) pairs -- against the
CPU oracle (`joint_future_pred(action_override=)`, held to the reference's own `forward(action_override=, mask_action_override=)` by
tools/fuzz_oracle_vs_reference.py) and against the FUSED rollout of the same case without overrides (bitwise).  Cases: tools/fuzz_cases.py.
usage: FUZZ_SEED=.. python tests/probes/gpu_fuzz_stepwise.py 40"""
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
from trafficbots_amd.runtime import teacher_forcing_mask  # noqa: E402
from trafficbots_amd.waymo_motion import WaymoMotion  # noqa: E402

torch.set_num_threads(min(8, torch.get_num_threads()))
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(os.environ.get("FUZZ_SEED", "313")))
fails, worst = 0, 0.0
for ci in range(n_cases):
    case = draw_case(rng)
    for k_ in ("rule_flags", "action_noise"):
        case.pop(k_, None)
    case["overrides"] = {k_: v for k_, v in case["overrides"].items() if k_.startswith("dynamics.")}  # (default history / teacher forcing here)
    sc, k, n_scene, step_end = case["scene"], case["k"], case["n_scene"], min(case["time_step_end"], 30)
    over = {"time_step_end": step_end, "n_joint_future": k, **case["overrides"]}
    cfg = load_model_config(overrides=over)
    sd = synth.case_state_dict(case)
    batch = synth.make_batch(case["base_seed"], n_scene, **sc)
    n, a = n_scene * k, sc["n_agent"]
    n_step = step_end - cfg["time_step_sim_start"] + 1
    eps = synth.make_latent_noise(case["base_seed"] + 99, n, a)
    ao_np, am_np = synth.make_action_override(case["base_seed"] + 55, n, a, n_step)
    ao, am = torch.from_numpy(ao_np).cuda(), torch.from_numpy(am_np).cuda()
    wm = WaymoMotion(**over)
    wm.load_state_dict(sd)
    fused = wm.test_step(batch, latent_eps=torch.from_numpy(eps).cuda(), generator=torch.Generator(device="cuda").manual_seed(case["base_seed"] % 2**31))
    gs = fused["goal_sample"].transpose(1, 2).reshape(n, a).contiguous()
    scene = wm.pre_processing(batch)
    scene.pop("gt", None)
    f = wm.model.encode_input_features(scene)
    latent = wm.model.latent_encoder()
    wm.model.goal_manager.pred_goal()
    latent.repeat_interleave_(k, 0)
    det = torch.zeros(n, a, dtype=torch.bool, device="cuda")
    det[::k] = True
    feats = dict(scene, map_feature=f["map_feature"], map_feature_valid=f["map_feature_valid"].to(torch.uint8), tl_feature=f["tl_feature"])
    gv = scene["agent_valid"].bool().any(1).repeat_interleave(k, 0)
    mask_tf = teacher_forcing_mask(scene["agent_valid"].bool())
    eps_t = torch.from_numpy(eps).cuda()

    def drive(with_override):
        wm.rollout(feats, latent, gs, gv, mask_tf, deterministic_latent=det, step_end=step_end, k_futures=k, latent_eps=eps_t, stepwise=True)
        for s_ in range(n_step):
            if with_override:
                wm.forward(action_override=ao[:, :, s_], mask_action_override=am[:, :, s_])
            else:
                wm.forward()
        torch.cuda.synchronize()
        b_ = wm.finish_rollout()
        b_.flatten_repeat(k)
        return b_

    msgs = []
    plain = drive(False)
    if not (torch.equal(plain.preds, fused["rollout_buffer"].preds) and torch.equal(plain.valid, fused["rollout_buffer"].valid)):
        msgs.append("stepwise without overrides is not the fused rollout, bit for bit")
    buf = drive(True)
    with torch.no_grad():
        r = Oracle(sd, cfg, torch.float32, hoist=True).joint_future_pred(batch, k, eps, step_end, dest_override=gs.cpu().numpy(),
                                                                        action_override=(ao_np, am_np))
    for key, got, ref in (("valid", buf.valid, r["valid"]), ("override", buf.override_masks, r["override_masks"]),
                          ("outside_map", buf.violations["outside_map"], r["outside_map"]), ("dest_reached", buf.violations["dest_reached"], r["dest_reached"])):
        if not (got.cpu().bool() == ref.bool()).all():
            msgs.append(f"{key} differs")
    v = r["valid"].bool()
    e_xy = float(((buf.preds.cpu() - r["preds"]).abs() * v.unsqueeze(-1))[..., :2].max())
    e_alp = float(((buf.action_log_probs.cpu() - r["action_log_probs"]).abs() * v).max())
    if e_xy > 1e-4 or e_alp > 2e-4:
        msgs.append("tolerance")
    moved = float((buf.preds - plain.preds).abs().max())  # (0 where nothing can move: scenes of type-less agents, dynamics.py masks their update)
    worst = max(worst, e_xy)
    fails += int(bool(msgs))
    print(f"case {ci:3d} B={n_scene} K={k} A={a:2d} P={sc['n_pl']:2d} T={sc['n_tl']:2d} S={step_end} edge={sc.get('edge', '-')} w={case.get('weight_mode', 'default')}: "
          f"overridden (agent, step) pairs {int(am_np.sum())} (moved the rollout by {moved:.1e} m), xy {e_xy:.1e} logp {e_alp:.1e}  {'ok' if not msgs else 'FAIL: ' + '; '.join(msgs)}", flush=True)
print(f"{n_cases} cases, {fails} failed; worst trajectory error {worst:.2e}")
sys.exit(1 if fails else 0)
