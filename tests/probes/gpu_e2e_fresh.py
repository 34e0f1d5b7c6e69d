"""Development probe (GPU box): WaymoMotion.test_step end to end with a FRESH host batch per call (four distinct numpy batches in
rotation, nothing resident), K = 1 and K = 6 at the headline shape, with check_range on / off; host-return vs GPU-inclusive time of
each stage at K = 1."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from trafficbots_amd import synth  # noqa: E402
from trafficbots_amd.waymo_motion import WaymoMotion  # noqa: E402

batches = [synth.make_batch(5000 + 37 * i, 32, n_agent=64, n_pl=256, n_tl=40) for i in range(4)]
sd = synth.make_state_dict(7)
for k in (1, 6):
    wm = WaymoMotion(time_step_end=90, n_joint_future=k)
    wm.load_state_dict(sd)
    for i in range(4):
        wm.test_step(batches[i % 4])
    torch.cuda.synchronize()
    for chk in (True, False):
        wm.check_range = chk
        n = 12
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            out = wm.test_step(batches[i % 4])
        torch.cuda.synchronize()
        t = (time.perf_counter() - t0) / n
        print(f"K={k} check_range={chk}: test_step {t * 1e3:.2f} ms per 32 scenes = {32 * 90 * k / t:.0f} scene-steps/s", flush=True)

    for chk in (True, False):  # the same stream of batches through the prefetcher
        wm.check_range = chk
        n = 16
        stream = [batches[i % 4] for i in range(n + 2)]
        it = iter(wm.prefetch(stream))
        wm.test_step(next(it))
        wm.test_step(next(it))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for sb in it:
            out = wm.test_step(sb)
        torch.cuda.synchronize()
        t = (time.perf_counter() - t0) / n
        print(f"K={k} check_range={chk} PREFETCH: test_step {t * 1e3:.2f} ms per 32 scenes = {32 * 90 * k / t:.0f} scene-steps/s", flush=True)

wm = WaymoMotion(time_step_end=90, n_joint_future=1)
wm.load_state_dict(sd)
wm.check_range = False
for i in range(4):
    wm.test_step(batches[i])
torch.cuda.synchronize()


def seg(name, fn, n=12):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        r = fn(i)
    t_host = (time.perf_counter() - t0) / n
    torch.cuda.synchronize()
    t_all = (time.perf_counter() - t0) / n
    print(f"{name:28s} host-return {t_host * 1e3:7.2f} ms   incl. GPU {t_all * 1e3:7.2f} ms", flush=True)
    return r


scene = seg("pre_processing", lambda i: wm.pre_processing(batches[i % 4]))
scene.pop("gt", None)
feats = seg("encode_input_features", lambda i: wm.model.encode_input_features(scene))
gv = scene["agent_valid"].bool().any(1)
seg("pred_goal", lambda i: wm.model.goal_manager.pred_goal())
seg("latent_encoder", lambda i: wm.model.latent_encoder())
buf, gs, glp = seg("joint_future_pred", lambda i: wm.joint_future_pred(scene, feats, wm.model.latent_encoder(), wm.model.goal_manager.pred_goal(), gv))
scores = torch.exp(buf.latent_log_probs[..., 0] + glp)
seg("waymo_post_processing", lambda i: wm.waymo_post_processing(valid=buf.valid[:, :, 0].any(-1), scores=scores, trajs=buf.preds[:, :, :, buf.step_future_start:], agent_type=scene["agent_type"]))
seg("test_step (whole)", lambda i: wm.test_step(batches[i % 4]))
