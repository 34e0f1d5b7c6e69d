"""Development probe (GPU box): HIP path vs the CPU oracle over randomly drawn small shapes and mask mixes (edge sizes: one agent,
one polyline, one traffic light, sizes around the 16 / 32 padding boundaries, K in 1..3).  Prints one line per case and a
summary; exits non-zero on the first mismatch.  Not part of the test-suite (the oracle makes it minutes long)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.trafficbots_oracle import Oracle  # noqa: E402
from trafficbots_amd import synth  # noqa: E402
from trafficbots_amd.config import load_model_config  # noqa: E402
from trafficbots_amd.waymo_motion import WaymoMotion  # noqa: E402

torch.set_num_threads(min(8, torch.get_num_threads()))  # (the CPU oracles: small matrices, see tests/probes/oracle_thread_timing.py)
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.default_rng(int(os.environ.get("FUZZ_SEED", "12345")))
EDGE_A = [1, 2, 15, 16, 17, 31, 32, 33, 48, 64, 65]
EDGE_P = [1, 2, 16, 31, 32, 33, 63, 64, 65, 96, 130]
EDGE_T = [1, 2, 15, 31, 32, 33, 40]
worst = 0.0
for ci in range(n_cases):
    a, p, t = int(rng.choice(EDGE_A)), int(rng.choice(EDGE_P)), int(rng.choice(EDGE_T))
    k, n_scene, step_end = int(rng.integers(1, 4)), int(rng.integers(1, 4)), int(rng.choice([12, 20, 30]))
    scene = dict(n_agent=a, n_pl=p, n_tl=t, p_invalid_agent=float(rng.choice([0.0, 0.3, 0.8])), p_late_spawn=float(rng.choice([0.0, 0.3])),
                 p_early_exit=float(rng.choice([0.0, 0.3])), p_invalid_pl=float(rng.choice([0.0, 0.4])),
                 p_invalid_node=float(rng.choice([0.0, 0.5])), p_tl_valid=float(rng.choice([0.0, 0.3, 1.0])),
                 pos_range=float(rng.choice([30.0, 100.0, 148.0])))
    seed = 20000 + ci
    cfg = load_model_config(overrides={"time_step_end": step_end, "n_joint_future": k})
    sd = synth.make_state_dict(seed)
    batch = synth.make_batch(seed, n_scene, **scene)
    eps = synth.make_latent_noise(seed + 1, n_scene * k, a)
    wm = WaymoMotion(time_step_end=step_end, n_joint_future=k)
    wm.load_state_dict(sd)
    out = wm.test_step(batch, latent_eps=torch.from_numpy(eps).cuda(), generator=torch.Generator(device="cuda").manual_seed(seed), tap_step=1)
    torch.cuda.synchronize()
    buf = out["rollout_buffer"]
    dest = out["goal_sample"].transpose(1, 2).reshape(n_scene * k, -1).cpu().numpy()
    with torch.no_grad():
        r = Oracle(sd, cfg, torch.float32, hoist=True).joint_future_pred(batch, k, eps, step_end, dest_override=dest, tap_steps=(1,))
    ok = True
    msgs = []
    e_map = float((out["input_feature_dict"]["map_feature"].cpu() - r["map_feature"]).abs().max())
    e_lat = float((out["latent_mean"].cpu() - r["latent_mean"]).abs().max())
    e_tap = float((buf.taps["tap_policy_feature"].cpu() - r["tap1/policy_feature"]).abs().max())
    for key, got, ref in (("valid", buf.valid, r["valid"]), ("override", buf.override_masks, r["override_masks"]),
                          ("outside_map", buf.violations["outside_map"], r["outside_map"]),
                          ("dest_reached", buf.violations["dest_reached"], r["dest_reached"])):
        if not (got.cpu() == ref).all():
            ok = False
            msgs.append(f"{key} differs")
    d = (buf.preds.cpu() - r["preds"]).abs() * r["valid"].unsqueeze(-1)
    e_xy = float(d[..., :2].max())
    if not np.isfinite(buf.preds.cpu().numpy()).all():
        ok = False
        msgs.append("non-finite preds")
    if e_map > 2e-5 or e_lat > 2e-5 or e_tap > 5e-6 or e_xy > 1e-4:
        ok = False
        msgs.append("tolerance")
    worst = max(worst, e_xy)
    print(f"case {ci:2d} A={a:3d} P={p:3d} T={t:2d} K={k} B={n_scene} S={step_end}  map {e_map:.1e} latent {e_lat:.1e} tap {e_tap:.1e} xy {e_xy:.1e}"
          f"  {'ok' if ok else 'FAIL: ' + ', '.join(msgs)}", flush=True)
    if not ok:
        print(json_dump := scene)
        sys.exit(1)
print(f"all {n_cases} cases ok; worst closed-loop xy error {worst:.2e}")
