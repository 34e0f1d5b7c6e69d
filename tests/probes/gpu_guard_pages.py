"""GPU probe, run by tests/test_gpu_boundary.py::test_no_access_outside_the_callers_buffers in a subprocess.  Every tensor the host
mirror hands to the C ABI is replaced (trafficbots_amd.hip.guard_hook) by a copy in a buffer that has UNMAPPED address space on both
sides (tests/guard/tb_guard.cpp: HIP virtual-memory calls), placed at the start of its mapping (GUARD_AT_END=0: an access a byte in
front of an array faults) or at its end (GUARD_AT_END=1: an access behind it faults), and copied back after the call.  An
out-of-bounds access of any kernel is then a GPU memory fault -- the process aborts -- instead of a silent touch of a neighbour.
Round 5 found such a read by accident (profiles/r05_experiments.txt item 16: the A-only launch of tb_rollout_begin read 8 bytes in
front of `preds`); a build with that bug put back (-DTB_DBG_OOB_STEP_INDEX) dies in this probe.  It walks the entry points a harness
uses -- test_step (fused rollout, K = 1 .. 3, sampled actions), validation_step, the stepwise rollout -- over a few shapes; the library's
own workspace and weight arena are ordinary allocations (not covered).  Prints GUARD-OK."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from trafficbots_amd import synth  # noqa: E402
from trafficbots_amd.runtime import teacher_forcing_mask  # noqa: E402
from trafficbots_amd.waymo_motion import WaymoMotion  # noqa: E402

sys.path.insert(0, os.path.join(ROOT, "tests", "guard"))
import guard_pool  # noqa: E402
from trafficbots_amd import hip  # noqa: E402

pool = guard_pool.install(os.environ.get("GUARD_AT_END", "0") == "1")
sd = synth.make_state_dict(7)
for i, (b, a, p, t, k) in enumerate(((2, 16, 32, 8, 1), (1, 33, 48, 5, 3), (3, 64, 66, 40, 2))):
    batch = synth.make_batch(7700 + i, b, n_agent=a, n_pl=p, n_tl=t, p_invalid_agent=0.2, p_late_spawn=0.2, p_invalid_pl=0.2, p_invalid_node=0.3)
    wm = WaymoMotion(time_step_end=30, n_joint_future=k)
    wm.load_state_dict(sd)
    g = torch.Generator(device="cuda").manual_seed(5 + i)
    out = wm.test_step(batch, generator=g)
    n = b * k
    eps = torch.from_numpy(synth.make_action_noise(31 + i, n, a, 30 - wm.hparams["time_step_sim_start"] + 1)).cuda() if hasattr(synth, "make_action_noise") else None
    if eps is not None:
        wm.test_step(batch, generator=g, action_eps=eps)
    wv = WaymoMotion(time_step_end=40, n_joint_future=k)  # (a validation batch carries its ground truth over the whole horizon)
    wv.load_state_dict(sd)
    wv.validation_step(synth.make_val_batch(7800 + i, b, n_agent=a, n_pl=p, n_tl=t), generator=g)
    # stepwise: begin / step / state, with a state override at one step
    scene = wm.pre_processing(batch)
    scene.pop("gt", None)
    f = wm.model.encode_input_features(scene)
    latent = wm.model.latent_encoder()
    goal = wm.model.goal_manager.pred_goal()
    latent.repeat_interleave_(k, 0)
    goal.repeat_interleave_(k, 0)
    det = torch.zeros(n, a, dtype=torch.bool, device="cuda")
    det[::k] = True
    gs = goal.sample(det, generator=g)
    feats = dict(scene, map_feature=f["map_feature"], map_feature_valid=f["map_feature_valid"].to(torch.uint8), tl_feature=f["tl_feature"])
    gv = scene["agent_valid"].bool().any(1).repeat_interleave(k, 0)
    wm.rollout(feats, latent, gs, gv, teacher_forcing_mask(scene["agent_valid"].bool()), deterministic_latent=det, step_end=30, k_futures=k,
               generator=g, stepwise=True)
    for _ in range(6):
        wm.forward()
    st = wm.engine.rollout_state()
    wm.finish_rollout()
    torch.cuda.synchronize()
    assert torch.isfinite(out["rollout_buffer"].preds).all() and torch.isfinite(st["agent_state"]).all()
# the other kernel families on one shape each: bf16 step kernels (also on the eight-wave assist carve, which TB_STEP_AW=2 forces for a
# launch of any size), the exact-fp32 kernels, K = 6 on the three-workgroups-per-CU carve's shape class (> 512 tiles)
batch = synth.make_batch(7900, 2, n_agent=33, n_pl=70, n_tl=9, p_invalid_agent=0.2, p_invalid_pl=0.2, p_invalid_node=0.3)
for prec, env in (("bf16", {}), ("bf16", {"TB_STEP_AW": "2"}), ("fp32_exact", {})):
    os.environ.update(env)
    wp = WaymoMotion(time_step_end=25, n_joint_future=2, operand_precision=prec)
    wp.load_state_dict(sd)
    o = wp.test_step(batch, generator=torch.Generator(device="cuda").manual_seed(9))
    torch.cuda.synchronize()
    assert torch.isfinite(o["rollout_buffer"].preds).all(), prec
    for k_ in env:
        del os.environ[k_]
big = synth.make_batch(7950, 32, n_agent=64, n_pl=64, n_tl=8)
wb = WaymoMotion(time_step_end=14, n_joint_future=6, operand_precision="bf16")  # 32 x 6 x 4 = 768 tiles
wb.load_state_dict(sd)
assert torch.isfinite(wb.test_step(big, generator=torch.Generator(device="cuda").manual_seed(10))["rollout_buffer"].preds).all()
torch.cuda.synchronize()
# the headline shape (128 tiles: helper workgroups + L2 warmers on the idle CUs) in the fp32-accurate kernels, and a stress-shaped scene
# (128 agents, 1024 polylines: the 32-block key walks; bf16 takes the assist-wave carve by itself at this polyline count)
head = synth.make_batch(5000, 32, n_agent=64, n_pl=256, n_tl=40)
wh = WaymoMotion(time_step_end=16, n_joint_future=1)
wh.load_state_dict(sd)
for _ in range(2):  # (the second call sees the same argument set again)
    oh = wh.test_step(head, generator=torch.Generator(device="cuda").manual_seed(11))
assert torch.isfinite(oh["rollout_buffer"].preds).all()
stress = synth.make_batch(5100, 2, n_agent=128, n_pl=1024, n_tl=40, p_invalid_pl=0.2)
for prec in ("fp32", "bf16"):
    ws = WaymoMotion(time_step_end=14, n_joint_future=1, operand_precision=prec)
    ws.load_state_dict(sd)
    assert torch.isfinite(ws.test_step(stress, generator=torch.Generator(device="cuda").manual_seed(12))["rollout_buffer"].preds).all(), prec
torch.cuda.synchronize()
guard_pool.uninstall(pool)
print(f"GUARD-OK ({pool.n_shadow} guarded buffers, at_end={pool.at_end}, granule {pool.lib.tbg_granularity()} B)")
