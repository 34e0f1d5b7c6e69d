"""Development probe (GPU box): HIP rollout vs the CPU oracle on seeded scenes, with per-step error
growth, plus a first timing at the headline shape.  Not part of the product or the test-suite."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.trafficbots_oracle import Oracle  # noqa: E402
from trafficbots_amd import synth  # noqa: E402
from trafficbots_amd.config import load_model_config  # noqa: E402
from trafficbots_amd.runtime import HipEngine, scene_from_batch  # noqa: E402


def compare(name, case, tap_step):
    cfg = load_model_config(overrides={"time_step_end": case["time_step_end"], "n_joint_future": case["k"]})
    sd = synth.make_state_dict(case["weight_seed"])
    batch = synth.make_batch(case["base_seed"], case["n_scene"], **case["scene"])
    k = case["k"]
    n = case["n_scene"] * k
    a = case["scene"]["n_agent"]
    eps = synth.make_latent_noise(case["base_seed"] + 99, n, a)
    orc = Oracle(sd, cfg, torch.float32, hoist=True)
    with torch.no_grad():
        ref = orc.joint_future_pred(batch, k, eps, case["time_step_end"], tap_steps=(tap_step,))
    inp, f = ref["_inp"], ref["_feats"]
    dev = torch.device("cuda:0")
    eng = HipEngine(cfg)
    eng.load_state_dict(sd)
    s = scene_from_batch(batch, dev)
    feats = {
        "map_feature": f["map_feature"].to(dev).contiguous(),
        "map_feature_valid": f["map_valid"].to(torch.uint8).to(dev).contiguous(),
        "tl_feature": f["tl_feature"].to(dev).contiguous(),
    }
    dest = ref["goal_sample"].transpose(1, 2).reshape(n, a)
    gv = inp["agent_valid"].any(1).repeat_interleave(k, 0)
    out = eng.rollout(s, feats, ref["latent_sample"].to(dev), ref["latent_mean"].to(dev), dest.to(dev), gv.to(dev), k,
                      case["time_step_end"], tap_step=tap_step)
    torch.cuda.synchronize()
    b = case["n_scene"]

    def unflat(x):  # [N,A,S,..] -> [B,A,K,S,..]
        return x.reshape(b, k, *x.shape[1:]).transpose(1, 2)

    print(f"== {name}: N={n} A={a} P={case['scene']['n_pl']} steps={case['time_step_end']}")
    for key in ("valid", "override_masks", "outside_map", "outside_map_this_step", "dest_reached", "dest_reached_this_step"):
        mine = unflat(out[key].cpu().bool())
        print(f"  {key:24s} mismatches = {(mine != ref[key]).sum().item()} / {mine.numel()}")
    pm, pr = unflat(out["preds"].cpu()), ref["preds"]
    vm = ref["valid"].unsqueeze(-1)
    d = ((pm - pr).abs() * vm)
    per_step = d[..., :2].amax(dim=(0, 1, 2, 4))
    print("  max|dxy| per step:", " ".join(f"{x:.1e}" for x in per_step.tolist()[:: max(1, len(per_step) // 15)]))
    print(f"  max|dxy| = {d[..., :2].max():.3e}  max|dyaw| = {d[..., 2].max():.3e}  max|dspd| = {d[..., 3].max():.3e}")
    print(f"  action_log_probs maxabs = {(unflat(out['action_log_probs'].cpu()) - ref['action_log_probs']).abs().max():.3e}")
    llp = out["latent_log_prob"].cpu().reshape(b, k, a).transpose(1, 2)
    print(f"  latent_log_prob maxabs = {(llp - ref['latent_log_probs'][..., 0]).abs().max():.3e}")
    tp = f"tap{tap_step}/"
    if tp + "policy_feature" in ref:
        print(f"  tap{tap_step} agent_feature maxabs = {(out['tap_agent_feature'].cpu() - ref[tp + 'agent_feature']).abs().max():.3e}")
        print(f"  tap{tap_step} policy_feature maxabs = {(out['tap_policy_feature'].cpu() - ref[tp + 'policy_feature']).abs().max():.3e}"
              f"  (ref absmax {ref[tp + 'policy_feature'].abs().max():.2f})")
    print(f"  final_hidden maxabs = {(out['final_hidden'].cpu().reshape(3, n * a, 128) - ref['final_hidden']).abs().max():.3e}")
    del eng


def timing():
    cfg = load_model_config()
    sd = synth.make_state_dict(7)
    b, a, p, t = 32, 64, 256, 40
    batch = synth.make_batch(5000, b, n_agent=a, n_pl=p, n_tl=t)
    dev = torch.device("cuda:0")
    eng = HipEngine(cfg)
    eng.load_state_dict(sd)
    s = scene_from_batch(batch, dev)
    g = torch.Generator(device="cpu").manual_seed(1)
    feats = {
        "map_feature": torch.randn(b, p, 128, generator=g).to(dev),
        "map_feature_valid": torch.ones(b, p, dtype=torch.uint8, device=dev),
        "tl_feature": torch.randn(b, 11, t, 128, generator=g).to(dev),
    }
    z = torch.randn(b, a, 16, generator=g).to(dev)
    dest = torch.randint(0, p, (b, a), generator=g).to(dev)
    gv = torch.ones(b, a, dtype=torch.uint8, device=dev)
    out = None
    eng.set_timing(True)
    for it in range(4):
        torch.cuda.synchronize()
        t0 = time.time()
        out = eng.rollout(s, feats, z, z * 0, dest, gv, 1, 90, out=out)
        torch.cuda.synchronize()
        dt_ = time.time() - t0
        tm = eng.get_timing()
        print(f"[timing] rollout B=32 A=64 P=256 90 steps: {dt_ * 1e3:.2f} ms -> {32 * 90 / dt_:.0f} scene-steps/s; "
              f"fused k_step {tm['fused_ms'] / max(1, tm['n_fused']) * 1e3:.1f} us  edge {tm['edge_ms']:.3f} ms  prologue {tm['prologue_ms']:.3f} ms")
    print("   finite:", bool(torch.isfinite(out["preds"]).all()), " valid frac:", out["valid"].float().mean().item())


if __name__ == "__main__":
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    cases = {
        "small_k1": dict(base_seed=2000, n_scene=2, k=1, weight_seed=7, time_step_end=90, scene=dict(n_agent=8, n_pl=32, n_tl=40)),
        "masks_k3": dict(base_seed=3000, n_scene=3, k=3, weight_seed=8, time_step_end=90,
                         scene=dict(n_agent=16, n_pl=48, n_tl=40, p_invalid_agent=0.3, p_late_spawn=0.3, p_early_exit=0.2,
                                    p_invalid_pl=0.2, p_invalid_node=0.5, pos_range=140.0)),
        "degenerate": dict(base_seed=4000, n_scene=2, k=2, weight_seed=8, time_step_end=40,
                           scene=dict(n_agent=16, n_pl=16, n_tl=40, p_invalid_agent=0.97, p_tl_valid=0.0, p_invalid_pl=0.5)),
        "headline_2": dict(base_seed=5000, n_scene=2, k=1, weight_seed=7, time_step_end=90, scene=dict(n_agent=64, n_pl=256, n_tl=40)),
    }
    which = sys.argv[1:] or list(cases.keys())
    for nm in which:
        if nm == "timing":
            timing()
        else:
            compare(nm, cases[nm], tap_step=1)
            compare(nm, cases[nm], tap_step=12)
