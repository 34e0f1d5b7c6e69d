"""Development probe (GPU box): per-step closed-loop distances of ONE case of gpu_fuzz_validation.py (FUZZ_SEED / FUZZ_ONLY as there):
oracle fp32 vs fp64 (the arithmetic's own noise), HIP vs fp64, HIP vs fp32, with the helper workgroups on and off and with the
fp32-MFMA twin kernel."""
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

only = int(os.environ["FUZZ_ONLY"])
rng = np.random.default_rng(int(os.environ.get("FUZZ_SEED", "777")))
EDGE_A = [1, 2, 15, 16, 17, 31, 33, 48, 65]
EDGE_P = [2, 16, 31, 33, 64, 65, 130]
EDGE_T = [1, 2, 15, 33, 40]
for ci in range(only + 1):
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
seed = 30000 + only
cfg = load_model_config(overrides=over)
sd = synth.make_state_dict(seed)
batch = synth.make_val_batch(seed, n_scene, **scene)
with torch.no_grad():
    r = Oracle(sd, cfg, torch.float32).reactive_replay(batch, step_end)
    r64 = Oracle(sd, cfg, torch.float64).reactive_replay(batch, step_end)
m = r["valid"].unsqueeze(-1).double()
noise = ((r["preds"].double() - r64["preds"]).abs() * m)[..., :2].amax((0, 1, 3))
print(scene, over)
print("step  ref32-vs-64 " + " ".join(f"{float(x):.1e}" for x in noise[9::10]))
for name, env in (("helpers on", {}), ("helpers off", {"TB_STEP_HELPERS": "0"}), ("fp32-MFMA twin", {"TB_STEP_KERNEL": "fp32"})):
    for k in ("TB_STEP_HELPERS", "TB_STEP_KERNEL"):
        os.environ.pop(k, None)
    os.environ.update(env)
    wm = WaymoMotion(**over)
    wm.load_state_dict(sd)
    buf = wm.validation_step(batch)["reactive_replay"]["rollout_buffer"]
    torch.cuda.synchronize()
    pr = buf.preds[:, :, 0].cpu().double()
    d64 = ((pr - r64["preds"]).abs() * m)[..., :2].amax((0, 1, 3))
    d32 = ((pr - r["preds"].double()).abs() * m)[..., :2].amax((0, 1, 3))
    cm = torch.cummax(noise, 0).values
    ok = bool((d64 <= torch.clamp(1.5 * cm, min=1e-4)).all() and (d32 <= torch.clamp(2.5 * cm, min=1e-4)).all())
    print(f"{name:15s} hip-vs-64  " + " ".join(f"{float(x):.1e}" for x in d64[9::10]) + f"   max {float(d64.max()):.2e}")
    print(f"{'':15s} hip-vs-32  " + " ".join(f"{float(x):.1e}" for x in d32[9::10]) + f"   max {float(d32.max()):.2e}  -> {'inside' if ok else 'OUTSIDE'}")
    print(f"{'':15s} flags equal: {bool((buf.valid[:, :, 0].cpu() == r['valid']).all())}")
