"""Development probe (GPU box): per-step closed-loop distances of ONE case of gpu_fuzz_validation.py (FUZZ_SEED / FUZZ_ONLY as there):
oracle fp32 vs fp64 (the arithmetic's own noise), an ensemble of FUZZ_MEMBERS (default 32) oracle fp32 runs on re-ordered batches
(tools/ensemble.py) with its per-step prediction bounds, and HIP vs fp64 / vs fp32 with the helper workgroups on and off and with
the fp32-MFMA twin kernel -- each with the step at which it first leaves a bound (if it does) and its rank among the members there."""
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
from tools import ensemble  # noqa: E402

n_mem = int(os.environ.get("FUZZ_MEMBERS", "32"))
r32p, r64p = r["preds"].numpy(), r64["preds"].numpy()
v32, v64 = r["valid"].numpy(), r["valid"].numpy() & r64["valid"].numpy()
base64 = ensemble.spread_per_step(r32p, r64p, v64, 2)
mem32, mem64 = [], []
with torch.no_grad():
    for mi in range(n_mem):
        pb, perm = ensemble.permute_batch({k_: np.asarray(v) for k_, v in batch.items()}, 7919 * (only + 1) + mi)
        rm = Oracle(sd, cfg, torch.float32, gemm_order_seed=4001 * (only + 1) + mi).reactive_replay(pb, step_end)
        mp, mv = perm.agents_back(rm["preds"].numpy()), perm.agents_back(rm["valid"].numpy())
        mem32.append(ensemble.spread_per_step(mp, r32p, mv & v32, 2))
        mem64.append(ensemble.spread_per_step(mp, r64p, mv & v64, 2))
mem32, mem64 = np.stack(mem32), np.stack([base64] + mem64)
cm32, cm64 = np.maximum.accumulate(mem32, 1), np.maximum.accumulate(mem64, 1)
fmt = lambda x: " ".join(f"{float(v):.1e}" for v in x[9::10])  # noqa: E731
print(scene, over)
print(f"steps 10, 20, ... ; {n_mem} members")
print(f"{'ref32 vs 64':28s} " + fmt(base64))
for first in (8, n_mem):
    q64, q32, cb = ensemble.prediction_bound(mem64[: first + 1]), ensemble.prediction_bound(mem32[:first]), np.maximum.accumulate(base64)
    b64, b32 = np.maximum(q64, cb + q32), np.maximum(q32, cb + q64)  # (the two triangles of tests/test_gpu_parity.py)
    print(f"{'bound vs 64, ' + str(first) + ' members':28s} " + fmt(b64))
    print(f"{'bound vs 32, ' + str(first) + ' members':28s} " + fmt(b32))
print(f"{'members vs 64: median':28s} " + fmt(np.median(cm64, 0)))
print(f"{'members vs 64: max':28s} " + fmt(cm64.max(0)))
print(f"{'members vs 32: max':28s} " + fmt(cm32.max(0)))
for name, env in (("helpers on", {}), ("helpers off", {"TB_STEP_HELPERS": "0"}), ("fp32-MFMA twin", {"TB_STEP_KERNEL": "fp32"})):
    for k in ("TB_STEP_HELPERS", "TB_STEP_KERNEL"):
        os.environ.pop(k, None)
    os.environ.update(env)
    wm = WaymoMotion(**over)
    wm.load_state_dict(sd)
    buf = wm.validation_step(batch)["reactive_replay"]["rollout_buffer"]
    torch.cuda.synchronize()
    hp = buf.preds[:, :, 0].cpu().numpy()
    d64 = np.maximum.accumulate(ensemble.spread_per_step(hp, r64p, v64, 2))
    d32 = np.maximum.accumulate(ensemble.spread_per_step(hp, r32p, v32, 2))
    print(f"{name:15s} {'hip vs 64':12s} " + fmt(d64) + f"   max {d64.max():.2e}")
    print(f"{'':15s} {'hip vs 32':12s} " + fmt(d32) + f"   max {d32.max():.2e}")
    for first in (8, n_mem):
        q64, q32, cb = ensemble.prediction_bound(mem64[: first + 1]), ensemble.prediction_bound(mem32[:first]), np.maximum.accumulate(base64)
        b64, b32 = np.maximum(1e-4, np.maximum(q64, cb + q32)), np.maximum(1e-4, np.maximum(q32, cb + q64))
        raw64 = ensemble.spread_per_step(hp, r64p, v64, 2)
        raw32 = ensemble.spread_per_step(hp, r32p, v32, 2)
        out64, out32 = np.nonzero(raw64 > b64)[0], np.nonzero(raw32 > b32)[0]
        msg = []
        for lab, o, raw, bb, cm in (("fp64", out64, raw64, b64, cm64[: first + 1]), ("fp32", out32, raw32, b32, cm32[:first])):
            if len(o):
                s0 = int(o[0])
                worst = int(o[np.argmax(raw[o] / bb[o])])
                msg.append(f"vs {lab}: {len(o)} steps outside, first at step {s0 + 1}, worst at step {worst + 1} ({raw[worst]:.2e} against {bb[worst]:.2e}; "
                           f"{int((cm[:, worst] >= raw[worst]).sum())}/{cm.shape[0]} members at least as far)")
        print(f"{'':15s} against the {first}-member bounds: " + ("inside" if not msg else "; ".join(msg)))
    if name == "helpers on":  # who deviates, and what happened to them before
        dev = (np.abs(hp.astype(np.float64) - r32p) * v32[..., None])[..., :2].max(-1)  # [B, A, S]
        ovr = r["override_masks"].numpy()
        for b_, a_ in sorted(((b_, a_) for b_ in range(dev.shape[0]) for a_ in range(dev.shape[1])), key=lambda t_: -dev[t_].max())[:4]:
            d_ = dev[b_, a_]
            first = int(np.argmax(d_ > 2e-5)) if (d_ > 2e-5).any() else -1
            vv = v32[b_, a_].astype(int)
            print(f"{'':15s} scene {b_} agent {a_}: max {d_.max():.2e}, first > 2e-5 at step {first + 1}; valid from step "
                  f"{int(np.argmax(vv)) + 1 if vv.any() else -1} to {len(vv) - int(np.argmax(vv[::-1]))}, teacher-forced steps "
                  f"{np.nonzero(ovr[b_, a_])[0][:3] + 1}..{np.nonzero(ovr[b_, a_])[0][-3:] + 1 if ovr[b_, a_].any() else ''}; "
                  f"|x|,|y| at the end {np.abs(r32p[b_, a_, -1, :2]).round(1)}, speed {abs(float(r32p[b_, a_, -1, 3])):.2f}; "
                  "dev at steps 40,45,..: " + " ".join(f"{x:.1e}" for x in d_[39::5]))
    print(f"{'':15s} end rank vs fp64 {ensemble.rank_among(mem64, float(d64[-1]))}, vs fp32 {ensemble.rank_among(mem32, float(d32[-1]))}; "
          f"flags equal: {bool((buf.valid[:, :, 0].cpu() == r['valid']).all())}")
