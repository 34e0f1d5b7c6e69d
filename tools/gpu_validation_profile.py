"""Timing of the validation path at the headline shape: 32 scenes x 64 agents x 256 polylines, 91-step ground truth.
Prints per-stage milliseconds (HIP events on the torch stream).  Run under rocprofv3 --kernel-trace --stats for the kernel table."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from trafficbots_amd import synth  # noqa: E402
from trafficbots_amd.runtime import teacher_forcing_mask  # noqa: E402
from trafficbots_amd.waymo_motion import WaymoMotion  # noqa: E402

K = int(os.environ.get("TB_K", "6"))
wm = WaymoMotion(time_step_end=90, n_joint_future=K)
wm.load_state_dict(synth.make_state_dict(7))
batch = synth.make_val_batch(5000, 32, n_agent=64, n_pl=256, n_tl=40, p_future_spawn=0.3, p_future_exit=0.2, p_invalid_agent=0.1)


def timed(fn, n=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        r = fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n, r


scene = wm.pre_processing(batch)
gt = scene["gt"]
t_enc, feats = timed(lambda: wm.model.encode_input_features(scene))
t_post, post = timed(lambda: wm.model.latent_encoder(posterior=True, gt=gt))
prior = wm.model.latent_encoder()
goal_gt, goal_valid = wm.model.goal_manager.get_gt_goal(scene["agent_valid"], gt.get("gt_goal"), gt["gt_dest"])
tf = wm.hparams["teacher_forcing_reactive_replay"]
mask_tf = teacher_forcing_mask(gt["agent_valid"].bool(), tf["step_spawn_agent"], tf["step_warm_start"])
t_rr, buf = timed(lambda: wm.reactive_replay(scene, feats, mask_tf, post, goal_gt, goal_valid))
gv, gs = wm._gt_slices(gt, 1, 90)
raw = {"valid": buf.valid, "preds": buf.preds, "override_masks": buf.override_masks}
t_train, _ = timed(lambda: wm.engine.train_partials(
    raw, gv, gs, scene["agent_size"], dest_logits=wm.model._enc["dest_logits"], goal_valid=goal_valid, gt_dest=goal_gt,
    post={"latent_mean": post.mean, "latent_valid": post.valid}, prior={"latent_mean": prior.mean, "latent_valid": prior.valid}))
t_val, out = timed(lambda: wm.validation_step(batch), n=3)
print(f"encode_scene {t_enc:.2f} ms | encode_posterior {t_post:.2f} ms | reactive_replay (rollout + reward + goal_reached) {t_rr:.2f} ms | "
      f"train_partials {t_train:.3f} ms | validation_step (replay + K={K} futures + metrics + post-processing, incl. host pre-processing) {t_val:.2f} ms")
print({k: round(v, 4) for k, v in wm.train_metrics_reactive_replay.compute().items()})
print({k: round(v, 4) for k, v in wm.err_metrics_joint_future_pred.compute().items()})
