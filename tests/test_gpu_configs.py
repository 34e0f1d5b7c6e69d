"""GPU tests of the BASELINE.json configurations beyond the headline (configs[3]: K = 6 futures with bf16 operands at B = 32;
configs[4]: the A = 128 / P = 1024 / 170-step stress shape, bf16 and fp32-accurate) and of the bench command's N > 1 path."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, golden_inputs, load_golden

pytestmark = pytest.mark.gpu

REPORT = {}


def _wm(**over):
    from trafficbots_amd.waymo_motion import WaymoMotion

    return WaymoMotion(**over)


def _per_step_max(a, b, valid):
    """max-abs xy difference per simulation step of two [B,A,K,S,4] buffers over `valid` [B,A,K,S]"""
    d = (a[..., :2] - b[..., :2]).abs() * valid.unsqueeze(-1)
    return d.amax(dim=(0, 1, 2, 4))


@pytest.mark.parametrize("shape", [dict(seed=4300, n_scene=3, n_agent=24, n_pl=48, n_tl=12, p_invalid_agent=0.2, p_late_spawn=0.2, pos_range=120.0),
                                   dict(seed=4400, n_scene=1, n_agent=64, n_pl=256, n_tl=40),
                                   # round 5: the shapes the bf16 kernels are sold on, past 60 steps -- BASELINE
                                   # configs[3] (K = 6 futures of one scene) and one scene of configs[4]'s stress shape; the latter also
                                   # on the eight-wave assist carve (TB_STEP_AW=2 forces it for a launch of any size)
                                   dict(seed=4500, n_scene=1, n_agent=64, n_pl=256, n_tl=40, k=6, step_end=70),
                                   dict(seed=4600, n_scene=1, n_agent=128, n_pl=1024, n_tl=40, k=1, step_end=70, also_aw=True,
                                        p_invalid_pl=0.3)])
def test_bf16_rollout_against_the_bf16_oracle(shape, monkeypatch):
    """VERDICT r03 task 1 (d): `operand_precision = "bf16"` had no accuracy bound past the teacher-forced steps.  The oracle's
    `operand_round="bf16"` mode rounds every matrix-product operand of the per-step policy to bf16 where the library does (fp32
    accumulate; the encoders stay fp32-accurate), and the HIP bf16 rollout is held to the SAME closed-loop rule as the fp32 path
    (tools/ensemble.py::closed_loop_rule) over 30 steps, K = 2 -- anchors: that oracle (base run) and its fp64-accumulate twin; yardstick:
    16 further bf16-oracle runs on permuted batches with re-ordered Linear sums.  bf16 arithmetic is ~300x noisier than fp32 and the
    loop amplifies it (two bf16 oracle runs end ~1e-2 m apart at step 30 where bf16 and fp32 end ~5e-2 m apart): the limits are what
    those members measure.  One-shot check on top: the policy feature of the first step against the fp64-accumulate twin is as close
    as the bf16 oracle's own fp32-accumulate run is (same rounding points), and far closer than bf16 is to fp32."""
    from oracle.trafficbots_oracle import Oracle
    from tools import ensemble
    from trafficbots_amd import synth
    from trafficbots_amd.config import load_model_config

    shape = dict(shape)
    seed, n_scene, k, step_end, also_aw = shape.pop("seed"), shape.pop("n_scene"), shape.pop("k", 2), shape.pop("step_end", 30), shape.pop("also_aw", False)
    sd = synth.make_state_dict(7)
    batch = synth.make_batch(seed, n_scene, **shape)
    a = shape["n_agent"]
    eps = synth.make_latent_noise(seed + 99, n_scene * k, a)
    cfg = load_model_config(overrides={"time_step_end": step_end, "n_joint_future": k})
    wm = _wm(time_step_end=step_end, n_joint_future=k, operand_precision="bf16")
    wm.load_state_dict(sd)
    with torch.no_grad():
        base = Oracle(sd, cfg, torch.float32, hoist=True, operand_round="bf16").joint_future_pred(batch, k, eps, step_end, tap_steps=(1,))
        dest = base["goal_sample"].numpy() if torch.is_tensor(base["goal_sample"]) else np.asarray(base["goal_sample"])
        dest = dest.reshape(n_scene, a, k).transpose(0, 2, 1).reshape(n_scene * k, a) if dest.ndim == 3 else dest
        twin = Oracle(sd, cfg, torch.float64, hoist=True, operand_round="bf16").joint_future_pred(batch, k, eps, step_end, dest_override=dest, tap_steps=(1,))
        f32 = Oracle(sd, cfg, torch.float32, hoist=True).joint_future_pred(batch, k, eps, step_end, dest_override=dest, tap_steps=(1,))
        members = []
        for i in range(16):
            pb, perm = ensemble.permute_batch({k_: np.asarray(v) for k_, v in batch.items()}, 911 * seed + i)
            r = Oracle(sd, cfg, torch.float32, hoist=True, operand_round="bf16", gemm_order_seed=17 * seed + i).joint_future_pred(
                pb, k, perm.agents_fwd(eps, k), step_end, dest_override=perm.dest_fwd(dest, k))
            members.append((perm.agents_back(r["preds"].numpy(), 1), perm.agents_back(r["valid"].numpy(), 1)))
    gs = torch.from_numpy(dest.reshape(n_scene, k, a).copy())
    out = wm.test_step(batch, latent_eps=torch.from_numpy(eps).cuda(), goal_sample=gs, tap_step=1)
    torch.cuda.synchronize()
    buf = out["rollout_buffer"]
    preds = buf.preds.cpu().numpy()
    assert (buf.valid.cpu().numpy() == base["valid"].numpy()).all()
    p32, p64, v = base["preds"].numpy(), twin["preds"].numpy(), base["valid"].numpy() & twin["valid"].numpy()
    ax = (0, 1, 2, 4)
    dist = lambda x, y, m: (np.abs(x.astype(np.float64) - y.astype(np.float64)) * m[..., None])[..., :2].max(axis=ax)  # noqa: E731
    ens32 = np.stack([dist(m, p32, mv & v) for m, mv in members])
    ens64 = np.stack([dist(p32, p64, v)] + [dist(m, p64, mv & v) for m, mv in members])
    r = ensemble.closed_loop_rule(dist(preds, p32, v), dist(preds, p64, v), ens32, ens64)
    vs_fp32 = dist(preds, f32["preds"].numpy(), v & f32["valid"].numpy())
    # one-shot: step-1 policy feature
    tap = buf.taps["tap_policy_feature"].cpu().double()
    vv = torch.from_numpy(base["valid"].numpy()[:, :, :, 0]).permute(0, 2, 1).reshape(n_scene * k, a, 1)
    e_hip = float(((tap - twin["tap1/policy_feature"]).abs() * vv).max())
    e_orc = float(((base["tap1/policy_feature"].double() - twin["tap1/policy_feature"]).abs() * vv).max())
    e_f32 = float(((f32["tap1/policy_feature"].double() - twin["tap1/policy_feature"]).abs() * vv).max())
    REPORT[f"bf16_vs_bf16_oracle/A{a}_K{k}_S{step_end}"] = dict({k_: v_ for k_, v_ in r.items() if k_ != "per_step"}, per_step=r["per_step"],
                                               bf16_vs_fp32_oracle_xy_final=float(np.maximum.accumulate(vs_fp32)[-1]),
                                               policy_feature_step1={"hip_vs_twin": e_hip, "bf16_oracle_vs_twin": e_orc, "fp32_oracle_vs_twin": e_f32})
    assert r["ok_vs_fp64"] and r["ok_vs_fp32"], {k_: v_ for k_, v_ in r.items() if k_ != "per_step"}
    # (the second condition: clearly closer to the twin than fp32 arithmetic is -- or, where bf16 and fp32 are only ~2x apart at this
    # step (the K = 6 case: 7.8e-4 / 1.7e-3), within 25 % of what the bf16 oracle's own fp32-accumulate run reaches)
    assert e_hip <= 3.0 * e_orc + 1e-5 and e_hip <= max(0.5 * e_f32, 1.25 * e_orc), (e_hip, e_orc, e_f32)
    if also_aw:  # the same inputs on the assist carve (tb::xba): another summation order of the same softmax, the same bound
        monkeypatch.setenv("TB_STEP_AW", "2")
        wm2 = _wm(time_step_end=step_end, n_joint_future=k, operand_precision="bf16")
        wm2.load_state_dict(sd)
        buf2 = wm2.test_step(batch, latent_eps=torch.from_numpy(eps).cuda(), goal_sample=gs, tap_step=1)["rollout_buffer"]
        torch.cuda.synchronize()
        monkeypatch.delenv("TB_STEP_AW")
        preds2 = buf2.preds.cpu().numpy()
        assert (buf2.valid.cpu().numpy() == base["valid"].numpy()).all() and not np.array_equal(preds2, preds)
        r2 = ensemble.closed_loop_rule(dist(preds2, p32, v), dist(preds2, p64, v), ens32, ens64)
        REPORT[f"bf16_vs_bf16_oracle/A{a}_assist_carve"] = {k_: v_ for k_, v_ in r2.items() if k_ != "per_step"}
        assert r2["ok_vs_fp64"] and r2["ok_vs_fp32"], {k_: v_ for k_, v_ in r2.items() if k_ != "per_step"}


def test_bf16_agent_alone_in_its_key_block():
    """the lean interaction walk (bf16 step kernels, `attention_walk_leanm_x`) starts a row tile's walk at a
    staggered key block; a row whose FIRST walked block holds no valid key but its own (hidden by MultiAgentTF's eye mask) kept the
    stand-in reference exponent, the next block's logits were rounded away against it and every valid key got the same weight --
    uniform attention, no error raised.  Here agents 0..31 and agent 40 are the valid ones (A = 64: agent 40's row tile starts at keys
    32..63, where only its own key is valid) and the attention in-projections are x3 ("sharp": near one-hot rows, so uniform weights
    are far from the softmax).  The step-1 policy feature of agent 40 must be as close to the bf16 oracle's fp64-accumulate twin as
    the bf16 oracle's own fp32 run is -- the bound of test_bf16_rollout_against_the_bf16_oracle, per agent."""
    from oracle.trafficbots_oracle import Oracle
    from trafficbots_amd import synth
    from trafficbots_amd.config import load_model_config

    n_scene, a, k, step_end = 2, 64, 1, 14
    sd = synth.make_state_dict(11, mode="sharp")
    batch = synth.make_batch(4700, n_scene, n_agent=a, n_pl=48, n_tl=12)
    keep = np.zeros(a, dtype=bool)
    keep[:32] = True
    keep[40] = True
    for key in list(batch.keys()):
        if key.endswith("agent/valid"):
            v = np.asarray(batch[key]).copy()
            ax = [i for i, n in enumerate(v.shape) if n == a][-1]
            shp = [1] * v.ndim
            shp[ax] = a
            batch[key] = v & keep.reshape(shp)
    eps = synth.make_latent_noise(4799, n_scene * k, a)
    cfg = load_model_config(overrides={"time_step_end": step_end, "n_joint_future": k})
    with torch.no_grad():
        base = Oracle(sd, cfg, torch.float32, hoist=True, operand_round="bf16").joint_future_pred(batch, k, eps, step_end, tap_steps=(1,))
        dest = np.asarray(base["goal_sample"]).reshape(n_scene * k, a)
        twin = Oracle(sd, cfg, torch.float64, hoist=True, operand_round="bf16").joint_future_pred(batch, k, eps, step_end, dest_override=dest, tap_steps=(1,))
    wm = _wm(time_step_end=step_end, n_joint_future=k, operand_precision="bf16")
    wm.load_state_dict(sd)
    gs = torch.from_numpy(dest.reshape(n_scene, k, a).copy())
    buf = wm.test_step(batch, latent_eps=torch.from_numpy(eps).cuda(), goal_sample=gs, tap_step=1)["rollout_buffer"]
    torch.cuda.synchronize()
    valid1 = base["valid"].numpy()[:, :, 0, 0]
    assert valid1[:, 40].all() and valid1[:, :32].all() and not valid1[:, 32:40].any() and not valid1[:, 41:].any(), "the case lost its shape"
    tap = buf.taps["tap_policy_feature"].cpu().double()
    e_hip = (tap - twin["tap1/policy_feature"]).abs().amax(-1)                       # [N, A]
    e_orc = (base["tap1/policy_feature"].double() - twin["tap1/policy_feature"]).abs().amax(-1)
    lone, rest = float(e_hip[:, 40].max()), float(e_hip[:, :32].max())
    REPORT["bf16_lone_agent_in_key_block"] = {"hip_vs_twin_agent40": lone, "hip_vs_twin_agents0_31": rest,
                                              "bf16_oracle_vs_twin_agent40": float(e_orc[:, 40].max()), "bf16_oracle_vs_twin_all": float(e_orc[:, keep].max())}
    bound = 3.0 * float(e_orc[:, keep].max()) + 1e-5
    assert lone <= bound and rest <= bound, REPORT["bf16_lone_agent_in_key_block"]
    # closed loop a few steps past the teacher-forced ones: same bookkeeping, bf16-sized distance to the bf16 oracle
    assert (buf.valid.cpu().numpy() == base["valid"].numpy()).all()
    d = np.abs(buf.preds.cpu().numpy() - base["preds"].numpy())[..., :2][base["valid"].numpy()]
    REPORT["bf16_lone_agent_in_key_block"]["xy_vs_bf16_oracle_step14"] = float(d.max())
    assert d.max() <= 2e-2, d.max()


def test_config3_k6_bf16_at_batch_32():
    """BASELINE configs[3] exactly: B = 32 scenes, K = 6 futures, A = 64, P = 256, 90 steps, operand_precision = "bf16", against
    the fp32-accurate run of the same inputs: teacher-forced bookkeeping equal (valid / override masks of the warm-up steps, the
    teacher-forced predictions within bf16's one-step error), everything finite, bitwise deterministic; the bf16 error is REPORTED
    per step (gpurun_out/configs_report.json), against the fp32-accurate HIP run and against the reference golden `headline_k6`
    (scene 0 of that golden's seed stream)."""
    from trafficbots_amd import synth

    sd = synth.make_state_dict(7)
    batch = synth.make_batch(6000, 32, n_agent=64, n_pl=256, n_tl=40)  # scene 0 = the `headline_k6` golden's scene
    g, meta = load_golden("headline_k6")
    eps = synth.make_latent_noise(meta["base_seed"] + 99 + 1000, 32 * 6, 64)
    eps[:6] = synth.make_latent_noise(meta["base_seed"] + 99, 6, 64)  # the golden's draws for scene 0
    eps_t = torch.from_numpy(eps).cuda()
    outs = {}
    wm = _wm(time_step_end=90, n_joint_future=6)
    wm.load_state_dict(sd)
    o = wm.test_step(batch, latent_eps=eps_t, generator=torch.Generator(device="cuda").manual_seed(5))
    goal = o["goal_sample"].clone()               # [B,A,K]; both precisions follow the same sampled destinations ...
    # ... and scene 0 follows the GOLDEN's destinations (round 6, VERDICT r05 weak #10: with the HIP sampler's own draws only 18 % of
    # scene 0's (agent, future) pairs had the golden's destination and the others perturbed everyone through the interaction block:
    # the reported 0.2 m "distance to the reference" at step 90 read like a parity failure and was none)
    goal[:1] = torch.from_numpy(g["goal_sample"]).to(goal)
    gs = goal.transpose(1, 2).contiguous().cpu()  # [B,K,A]
    o = wm.test_step(batch, latent_eps=eps_t, goal_sample=gs)
    outs["fp32"] = o["rollout_buffer"]
    wm = _wm(time_step_end=90, n_joint_future=6, operand_precision="bf16")
    wm.load_state_dict(sd)
    o = wm.test_step(batch, latent_eps=eps_t, goal_sample=gs)
    o2 = wm.test_step(batch, latent_eps=eps_t, goal_sample=gs)
    assert torch.equal(o["rollout_buffer"].preds, o2["rollout_buffer"].preds), "bf16 run is not bitwise deterministic"
    outs["bf16"] = o["rollout_buffer"]
    torch.cuda.synchronize()
    a, b = outs["fp32"], outs["bf16"]
    assert torch.isfinite(b.preds).all() and torch.isfinite(a.preds).all()
    assert a.preds.shape == (32, 64, 6, 90, 4)
    # teacher-forced steps 1..10: identical bookkeeping, predictions one bf16 policy step away from the fp32 ones
    assert torch.equal(a.override_masks, b.override_masks)
    assert torch.equal(a.valid[..., :11], b.valid[..., :11])
    per_step = _per_step_max(a.preds, b.preds, a.valid & b.valid)
    assert per_step[:10].max() <= 5e-2, per_step[:10]
    # flags: the discrete outputs may only differ where a trajectory moved (reported), never at the forced steps
    for key in ("outside_map", "dest_reached"):
        assert torch.equal(a.violations[key][..., :10], b.violations[key][..., :10]), key
    rep = {"bf16_vs_fp32_xy_per_step": [float(x) for x in per_step.cpu()],
           "valid_mismatch_frac": float((a.valid != b.valid).float().mean()),
           "dest_reached_mismatch_frac": float((a.violations["dest_reached"] != b.violations["dest_reached"]).float().mean())}
    # scene 0 against the reference golden (fp32 HIP run: the closed-loop envelope; bf16: reported)
    gp, gv = torch.from_numpy(g["preds"]).cuda(), torch.from_numpy(g["valid"]).cuda()
    # (scene 0 ran with the golden's destinations and latent draws: the whole trajectory is comparable, and the fp32-accurate run is
    # held to the closed-loop rule of the parity tests on it -- scene 0 inside a 32-scene, 1152-tile LEAN launch)
    from tools import ensemble

    for t in (11, 30, 60, 90):
        mt = gv[..., t - 1, None]
        rep[f"scene0_fp32_vs_reference_xy_step{t}"] = float(((a.preds[:1] - gp)[..., t - 1, :2].abs() * mt).max())
        rep[f"scene0_bf16_vs_reference_xy_step{t}"] = float(((b.preds[:1] - gp)[..., t - 1, :2].abs() * mt).max())
    p0 = a.preds[:1].cpu().numpy()
    both = (g["valid"] & g["valid_fp64"])[..., None]
    d32 = (np.abs(p0 - g["preds"]) * g["valid"][..., None])[..., :2].max(axis=(0, 1, 2, 4))
    d64 = (np.abs(p0.astype(np.float64) - g["preds_fp64"]) * both)[..., :2].max(axis=(0, 1, 2, 4))
    e = np.load(os.path.join(ROOT, "tests", "golden", "ensg", "headline_k6.npz"))
    r = ensemble.closed_loop_rule(d32, d64, e["ensg_d32"], e["ensg_d64"], 60)
    rep["scene0_fp32_rule"] = {k_: v_ for k_, v_ in r.items() if k_ != "per_step"}
    REPORT["config3_k6_bf16"] = rep
    assert rep["scene0_fp32_vs_reference_xy_step11"] <= 1e-4
    assert r["ok"], rep["scene0_fp32_rule"]


@pytest.mark.parametrize("prec", ["bf16", "fp32"])
def test_lean_carve_two_workgroups_per_cu_is_bitwise_identical(prec, monkeypatch):
    """A launch of more than 256 tiles runs the LEAN carve of the step kernel (k_step_x<false, true>: no goal / latent /
    LayerNorm-parameter tiles in LDS -- with fp16-pair planes no destination geometry either, and the kernel compiled for 256 VGPRs --
    i.e. < 80 KB), so that two workgroups share a CU.  It must not change a bit:
    TB_STEP_LEAN=0 (the full carve at every size) against the default, at 288 tiles (12 scenes x 6 futures x 4 row tiles), and a
    sub-batch that stays below the threshold (full carve) against its slice of the big batch (lean carve)."""
    from trafficbots_amd import synth

    sd = synth.make_state_dict(7)
    batch = synth.make_batch(8600, 12, n_agent=64, n_pl=96, n_tl=20, p_invalid_agent=0.1, p_late_spawn=0.2)
    eps = torch.from_numpy(synth.make_latent_noise(8601, 12 * 6, 64)).cuda()
    gen = torch.Generator(device="cuda")
    outs = {}
    for name, flag in (("lean", "1"), ("full", "0")):
        monkeypatch.setenv("TB_STEP_LEAN", flag)
        wm = _wm(time_step_end=40, n_joint_future=6, operand_precision=prec)
        wm.load_state_dict(sd)
        gen.manual_seed(5)
        outs[name] = wm.test_step(batch, latent_eps=eps, generator=gen)
    torch.cuda.synchronize()
    a, b = outs["lean"]["rollout_buffer"], outs["full"]["rollout_buffer"]
    assert torch.isfinite(a.preds).all()
    assert torch.equal(a.preds, b.preds), float((a.preds - b.preds).abs().max())
    assert torch.equal(a.valid, b.valid) and torch.equal(a.action_log_probs, b.action_log_probs)
    # 3 scenes x 6 x 4 = 72 tiles: the full carve (with helper workgroups); same scenes, same destinations
    monkeypatch.delenv("TB_STEP_LEAN")
    sub = {k: v[:3] for k, v in batch.items()}
    wm = _wm(time_step_end=40, n_joint_future=6, operand_precision=prec)
    wm.load_state_dict(sd)
    gs = outs["lean"]["goal_sample"][:3].transpose(1, 2).contiguous()  # [B,A,K] -> [B,K,A]
    c = wm.test_step(sub, latent_eps=eps[: 3 * 6], goal_sample=gs.cpu())["rollout_buffer"]
    assert torch.equal(c.preds, a.preds[:3]), float((c.preds - a.preds[:3]).abs().max())
    # one-tile instances, most of them with a single valid agent (interaction bypass) or none, no lit traffic light: 44 x 7 = 308 tiles
    batch2 = synth.make_batch(8700, 44, n_agent=9, n_pl=20, n_tl=6, p_invalid_agent=0.85, p_tl_valid=0.0)
    eps2 = torch.from_numpy(synth.make_latent_noise(8701, 44 * 7, 9)).cuda()
    res = {}
    for name, flag in (("lean", "1"), ("full", "0")):
        monkeypatch.setenv("TB_STEP_LEAN", flag)
        wm = _wm(time_step_end=30, n_joint_future=7, operand_precision=prec)
        wm.load_state_dict(sd)
        gen.manual_seed(6)
        res[name] = wm.test_step(batch2, latent_eps=eps2, generator=gen)["rollout_buffer"]
    assert torch.equal(res["lean"].preds, res["full"].preds) and torch.equal(res["lean"].valid, res["full"].valid)


def test_w3_carve_three_workgroups_per_cu_is_bitwise_identical(monkeypatch):
    """bf16 launches of more than 512 row tiles (BASELINE configs[3]: 32 scenes x 6 futures x 4 row tiles = 768) run the THIRD build of
    the step kernel (tb::xb3, tb_stepx_bf16w3_kernels.hip): weight units loaded by the consuming GEMM instead of one stage ahead
    (<= 168 VGPRs), GRU hidden state read from the rollout workspace instead of LDS copies (49 KB) -- three workgroups per CU, one
    dispatch round.  Same arithmetic in the same order: TB_STEP_W3=0 (the two-per-CU carve) gives the same bits, with and without the
    batched warm start (its A-half launch takes the same carve), on masks / late spawns and on one-tile instances with bypasses."""
    from trafficbots_amd import synth

    sd = synth.make_state_dict(7)
    gen = torch.Generator(device="cuda")
    cases = [
        (synth.make_batch(8800, 24, n_agent=64, n_pl=96, n_tl=20, p_invalid_agent=0.1, p_late_spawn=0.2), 6, 64, 30),   # 576 tiles, no warm start
        (synth.make_batch(8810, 32, n_agent=64, n_pl=64, n_tl=20), 6, 64, 24),                                            # 768 tiles, batched warm start
        (synth.make_batch(8820, 80, n_agent=9, n_pl=20, n_tl=6, p_invalid_agent=0.85, p_tl_valid=0.0), 7, 9, 20),        # 560 one-tile instances
        # 576 one-tile instances, two or three valid agents each, close to a tight map boundary (agents leave, instances fall back to
        # the single-agent bypass mid-rollout)
        (synth.make_batch(8830, 96, n_agent=9, n_pl=20, n_tl=6, p_invalid_agent=0.7, pos_range=58.0, boundary=60.0), 6, 9, 40),
    ]
    for batch, k, a, step_end in cases:
        n = batch["map/valid"].shape[0] * k
        eps = torch.from_numpy(synth.make_latent_noise(8801, n, a)).cuda()
        outs = {}
        for name, flag in (("w3", "1"), ("lean", "0")):
            monkeypatch.setenv("TB_STEP_W3", flag)
            wm = _wm(time_step_end=step_end, n_joint_future=k, operand_precision="bf16")
            wm.load_state_dict(sd)
            gen.manual_seed(5)
            outs[name] = wm.test_step(batch, latent_eps=eps, generator=gen)["rollout_buffer"]
        torch.cuda.synchronize()
        y = outs["lean"]
        for name in ("w3",):
            x = outs[name]
            assert torch.isfinite(x.preds).all()
            assert torch.equal(x.preds, y.preds), (name, k, float((x.preds - y.preds).abs().max()))
            assert torch.equal(x.valid, y.valid) and torch.equal(x.action_log_probs, y.action_log_probs)
            assert torch.equal(x.final["final_hidden"], y.final["final_hidden"])


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_config4_stress_shape_170_steps(prec):
    """BASELINE configs[4] shape at its full horizon: A = 128, P = 1024, time_step_end = 170 (160 future steps), 4 scenes, both
    operand precisions.  Size-independent properties (no oracle finishes this in seconds): finite, bitwise deterministic, sub-batch
    independence (scenes 1..2 alone = the same rows), the teacher-forced steps reproduce the history, yaw is never wrapped, agents
    that leave the map are killed; plus a short-horizon check against the CPU oracle (fp32 only: 1 scene, 14 steps)."""
    from trafficbots_amd import synth

    sd = synth.make_state_dict(7)
    scene = dict(n_agent=128, n_pl=1024, n_tl=40)
    batch = synth.make_batch(9400, 4, **scene)
    eps = torch.from_numpy(synth.make_latent_noise(9401, 4, 128)).cuda()
    wm = _wm(time_step_end=170, n_joint_future=1, operand_precision=prec)
    wm.load_state_dict(sd)
    a = wm.test_step(batch, latent_eps=eps)["rollout_buffer"]
    b = wm.test_step(batch, latent_eps=eps)["rollout_buffer"]
    torch.cuda.synchronize()
    assert a.preds.shape == (4, 128, 1, 170, 4)
    assert torch.isfinite(a.preds).all()
    assert torch.equal(a.preds, b.preds) and torch.equal(a.valid, b.valid), "not bitwise deterministic"
    sub = {k_: v[1:3] for k_, v in batch.items()}
    c = wm.test_step(sub, latent_eps=eps[1:3])["rollout_buffer"]
    assert torch.equal(c.preds, a.preds[1:3]), "a scene's result depends on its neighbours in the batch"
    hist = torch.from_numpy(batch["history/agent/pos"]).cuda()  # [B,11,A,2]
    gap = (a.preds[:, :, 0, :10, :2] - hist[:, 1:11].transpose(1, 2)).norm(dim=-1)
    assert gap.max() < 3.0
    # an agent flagged outside the map without ground truth behind it is invalid from the next step on (dynamics.py:161-167)
    out_this = a.violations["outside_map_this_step"][:, :, 0, 10:-1]
    nxt_valid = a.valid[:, :, 0, 11:]
    assert not (out_this & nxt_valid).any()
    REPORT[f"config4_stress_{prec}"] = {"valid_frac_final": float(a.valid[..., -1].float().mean()),
                                        "dest_reached_final": float(a.violations["dest_reached"][..., -1].float().mean()),
                                        "max_abs_xy": float(a.preds[..., :2].abs().max())}
    if prec == "bf16":
        wm32 = _wm(time_step_end=170, n_joint_future=1)
        wm32.load_state_dict(sd)
        r = wm32.test_step(batch, latent_eps=eps)["rollout_buffer"]
        per_step = _per_step_max(r.preds, a.preds, r.valid & a.valid)
        REPORT["config4_stress_bf16"]["bf16_vs_fp32_xy_per_step"] = [float(x) for x in per_step.cpu()]
        assert torch.equal(r.override_masks, a.override_masks)
        assert per_step[:10].max() <= 5e-2
        return
    from oracle.trafficbots_oracle import Oracle
    from trafficbots_amd.config import load_model_config

    cfg = load_model_config(overrides={"time_step_end": 14, "n_joint_future": 1})
    one = {k_: v[:1] for k_, v in batch.items()}
    wm14 = _wm(time_step_end=14, n_joint_future=1)
    wm14.load_state_dict(sd)
    o = wm14.test_step(one, latent_eps=eps[:1])
    dest = o["goal_sample"].transpose(1, 2).reshape(1, -1).cpu().numpy()
    with torch.no_grad():
        ref = Oracle(sd, cfg, torch.float32, hoist=True).joint_future_pred(one, 1, eps[:1].cpu().numpy(), 14, dest_override=dest)
    buf = o["rollout_buffer"]
    assert (buf.valid.cpu() == ref["valid"]).all()
    err = ((buf.preds.cpu() - ref["preds"]).abs() * ref["valid"].unsqueeze(-1))[..., :2].max().item()
    REPORT["config4_stress_fp32"]["xy_vs_oracle_14_steps"] = err
    assert err <= 1e-4
    # the 170-step run and the 14-step run agree bitwise on their common steps (nothing depends on the horizon)
    assert torch.equal(buf.preds[0, :, 0], a.preds[0, :, 0, :14])


def test_config4_full_per_gpu_batch_properties(monkeypatch):
    """BASELINE configs[4] at its full per-GPU load: 32 scenes x 128 agents x 1024 polylines x 170 steps, bf16
    -- what each of the 8 ranks runs.  No oracle finishes this; asserted: finite, bitwise deterministic, the 4-scene run of
    a 4-scene run of the same scenes is its first four rows (a scene does not see its neighbours at any batch size), kill flags
    consistent.
    Round 5: this launch (256 row tiles, 1024 polylines) takes the EIGHT-WAVE carve (assist waves, tb::xba): deterministic, but the two
    halves of a key walk are merged in another order than the four-wave kernel sums them, so the sub-batch identity is asserted with
    TB_STEP_AW=0 and the assist carve is held to it within bf16 rounding at the first free steps."""
    from trafficbots_amd import synth

    sd = synth.make_state_dict(7)
    scene = dict(n_agent=128, n_pl=1024, n_tl=40)
    batch = synth.make_batch(9400, 32, **scene)
    eps = torch.from_numpy(synth.make_latent_noise(9402, 32, 128)).cuda()
    wm = _wm(time_step_end=170, n_joint_future=1, operand_precision="bf16")
    wm.load_state_dict(sd)
    a = wm.test_step(batch, latent_eps=eps)["rollout_buffer"]
    b = wm.test_step(batch, latent_eps=eps)["rollout_buffer"]
    torch.cuda.synchronize()
    assert a.preds.shape == (32, 128, 1, 170, 4) and torch.isfinite(a.preds).all()
    assert torch.equal(a.preds, b.preds) and torch.equal(a.valid, b.valid)
    sub = {k_: v[:4] for k_, v in batch.items()}
    c = wm.test_step(sub, latent_eps=eps[:4])["rollout_buffer"]
    monkeypatch.setenv("TB_STEP_AW", "0")
    a4 = wm.test_step(batch, latent_eps=eps)["rollout_buffer"]  # the same batch on the four-wave kernel
    monkeypatch.delenv("TB_STEP_AW")
    assert torch.equal(c.preds, a4.preds[:4]) and torch.equal(c.valid, a4.valid[:4])
    assert not torch.equal(a.preds, a4.preds), "the full batch was expected to run on the assist carve (tb::xba)"
    both = (a.valid & a4.valid)[..., :12].unsqueeze(-1)
    d = ((a.preds - a4.preds).abs()[..., :12, :2] * both).amax(dim=(0, 1, 2, 4))
    assert torch.equal(a.valid[..., :12], a4.valid[..., :12]) and float(d[:10].max()) == 0.0 and float(d[10:12].max()) < 2e-3, d
    out_this = a.violations["outside_map_this_step"][:, :, 0, 10:-1]
    assert not (out_this & a.valid[:, :, 0, 11:]).any()
    assert bool(a.violations["dest_reached"][..., -1].any()) and float(a.valid[..., -1].float().mean()) > 0.3
    REPORT["config4_full_batch_bf16"] = {"valid_frac_final": float(a.valid[..., -1].float().mean()),
                                         "dest_reached_final": float(a.violations["dest_reached"][..., -1].float().mean())}


def _run_bench(extra, env_extra=None):
    env = dict(os.environ)
    env.update(env_extra or {})
    env.pop("RANK", None)
    env.pop("WORLD_SIZE", None)
    env.pop("LOCAL_RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline"] + extra,
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_plain_command_spawns_its_ranks():
    """`python bench.py --gpus 2` as a PLAIN command (no torchrun, no WORLD_SIZE) spawns its own two ranks and prints one line;
    with TB_BENCH_BACKEND=gloo the ranks share the visible GPU(s), which exercises shards, barriers, the max-over-ranks time and the
    single packed all-reduce on a 1-GPU box.  The N = 2 line covers twice the scenes of the N = 1 line of the same command."""
    one = _run_bench(["--gpus", "1", "--lean"])
    two = _run_bench(["--gpus", "2", "--lean"], {"TB_BENCH_BACKEND": "gloo"})
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2
    assert two["checks"]["scene_steps"] == 2 * one["checks"]["scene_steps"] == 2 * 32 * 90
    assert two["checks"]["finite"] and one["checks"]["finite"]
    # rank 1 rolled out scenes 32..63 of the seeded stream: different scenes, same count of agent slots
    assert two["reference_metric_states"]["counter_agent"] == 2 * one["reference_metric_states"]["counter_agent"]
    assert two["checks"]["valid_agent_steps"] != 2 * one["checks"]["valid_agent_steps"]
    for ln in (one, two):
        for key in ("metric", "value", "unit", "ms_per_step", "scaling", "roofline", "config"):
            assert key in ln
        assert ln["roofline"]["frac"] > 0 and ln["value"] > 0
    REPORT["bench_gloo_dry_run"] = {"n1_value": one["value"], "n2_value_two_ranks_one_gpu": two["value"]}


def test_bench_eight_ranks_share_the_device():
    """The driver's `bench.py --gpus 8` flow with all eight ranks on the one visible GPU (TB_BENCH_BACKEND=gloo, --lean): what an
    8-GPU node will run, minus RCCL and the peers -- eight shards of the seeded scene stream, ONE packed all-reduce for the metric
    reduction, none inside the timed passes, every rank's elapsed time in the line (VERDICT r04 task 7)."""
    ln = _run_bench(["--gpus", "8", "--lean", "--steps", "3", "--warmup", "1"], {"TB_BENCH_BACKEND": "gloo"})
    assert ln["n_gpus"] == 8 and ln["checks"]["finite"]
    assert ln["checks"]["scene_steps"] == 8 * 32 * 90
    r = ln["ranks"]
    assert r["ranks_seen"] == 8 and r["collectives_in_timed_passes"] == 0 and len(r["elapsed_s"]) == 8 and min(r["elapsed_s"]) > 0
    assert ln["reference_metric_states"]["counter_agent"] > 0 and ln["scaling"] == "weak"
    REPORT["bench_gloo_eight_ranks_one_gpu"] = {"value": ln["value"], "elapsed_s": r["elapsed_s"]}


def test_bench_n_ranks_carry_the_end_to_end_block():
    """VERDICT r05 task 8: the N > 1 line also measures the DROP-IN path on every rank at once (`tools/e2e_bench.measure_rank`: plain
    `test_step` calls and the lane pipeline, each rank its own host batches, its own stager and its own launching thread), gathered to
    rank 0 -- host-side contention shows before a multi-GPU node exists.  Two ranks share the visible GPU here (gloo)."""
    ln = _run_bench(["--gpus", "2", "--configs", "--sustain-seconds", "0"], {"TB_BENCH_BACKEND": "gloo"})
    assert ln["n_gpus"] == 2 and ln["ranks"]["ranks_seen"] == 2 and ln["ranks"]["collectives_in_timed_passes"] == 0
    e = ln["e2e"]
    per = e["per_rank"]
    assert len(per) == 2 and all("plain_ms_per_batch" in r and "pipeline_2_lanes_ms_per_batch" in r for r in per), e
    assert all(0 < r["pipeline_2_lanes_ms_per_batch"] and 0 < r["plain_ms_per_batch"] < 100 for r in per)
    REPORT["bench_gloo_two_ranks_e2e"] = per


def test_bench_one_rank_through_rccl():
    """The driver's multi-GPU command line with ONE rank on the real backend: `python -m torch.distributed.run --nproc-per-node 1
    bench.py --gpus 1` with TB_BENCH_FORCE_DIST=1 initialises the RCCL process group (backend "nccl", device_id), runs the barriers
    around the timed passes and the ONE packed float64 SUM all-reduce of trafficbots_amd/shard.py on the device -- the same code the
    N = 2, 4, 8 runs take (scene shards, per-rank tails), minus the peers a 1-GPU box does not have.  The line must equal the plain
    single-process run in everything but time."""
    env = dict(os.environ, TB_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "TB_BENCH_BACKEND"):
        env.pop(k, None)
    import socket

    with socket.socket() as sk:  # a free rendezvous port (a fixed one can still be in TIME_WAIT from an earlier run on the same box)
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline", "--lean"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    ln = json.loads(lines[0])
    plain = _run_bench(["--gpus", "1", "--lean"])
    assert ln["n_gpus"] == 1 and ln["ranks"]["ranks_seen"] == 1
    assert ln["ranks"]["collectives_for_the_metric_reduction"] == 1 and ln["ranks"]["collectives_in_timed_passes"] == 0
    assert plain["ranks"]["collectives_for_the_metric_reduction"] == 0  # (no process group: nothing to reduce)
    assert ln["checks"] == plain["checks"] and ln["reference_metric_states"] == plain["reference_metric_states"]
    REPORT["bench_one_rank_rccl"] = {"value": ln["value"], "plain_value": plain["value"], "ranks": ln["ranks"]}


def test_rollout_graph_replay_is_bitwise_identical(monkeypatch):
    """tb_rollout captures ONE hipGraph per rollout the second time it sees the same argument set (same buffers, sizes, switches) and
    replays it afterwards (the launching thread spends microseconds per rollout instead of ~70 us per launch -- eight ranks share
    the host's cores).  Replays, the capture pass, the plain passes and TB_ROLLOUT_GRAPH=0 give the same bits; new buffers or another
    step count fall back to plain launches and re-capture."""
    import bench
    from trafficbots_amd import synth
    from trafficbots_amd.config import load_model_config

    sd = synth.make_state_dict(7)
    res = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("TB_ROLLOUT_GRAPH", mode)
        for prec, n_scene, k in (("fp32", 6, 1), ("bf16", 4, 3)):
            cfg = load_model_config(overrides={"time_step_end": 30, "n_joint_future": k, "operand_precision": prec})
            c = bench.setup_case(cfg, sd, torch.device("cuda", 0), 0, n_scene, 64, 96, k, seed=8900)
            eng, scene, feats, z, enc, dest, gv = c["eng"], c["scene"], c["feats"], c["z"], c["enc"], c["dest"], c["gv"]
            out = None
            snaps = []
            for i in range(5):  # plain, capture, replay, replay, replay
                out = eng.rollout(scene, feats, z, enc["latent_mean"], dest, gv, k, 30, out=out)
                snaps.append(out["preds"].clone())
            other = eng.rollout(scene, feats, z, enc["latent_mean"], dest, gv, k, 30)          # fresh buffers: plain launches
            short = eng.rollout(scene, feats, z, enc["latent_mean"], dest, gv, k, 20)          # another horizon
            again = eng.rollout(scene, feats, z, enc["latent_mean"], dest, gv, k, 30, out=out)  # the first set once more
            torch.cuda.synchronize()
            eng.check_status()
            gs = eng.graph_stats()
            assert (gs["captured"] >= 1 and gs["replayed"] >= 3) if mode == "1" else gs == {"captured": 0, "replayed": 0}, (mode, gs)
            for sn in snaps[1:]:
                assert torch.equal(sn, snaps[0])
            assert torch.equal(other["preds"], snaps[0]) and torch.equal(again["preds"], snaps[0])
            assert torch.equal(short["preds"], snaps[0][:, :, :20])
            res[(mode, prec)] = snaps[0]
    for prec in ("fp32", "bf16"):
        assert torch.equal(res[("1", prec)], res[("0", prec)])


def test_encode_side_stream_is_bitwise_identical(monkeypatch):
    """tb_encode_scene runs the agent / traffic-light token encoders and the destination predictor's GRU scan on a side stream beside
    the map encoder (forked from / joined into the caller's stream with events); TB_ENCODE_SIDE=0 keeps everything on the caller's
    stream.  Same kernels, same arguments: every product of the encoder is bit-identical, call after call (the side stream is
    joined before the personality branch, so a second call cannot overtake the first one's workspace).  Round 5: the side stream also
    carries the latent branch's gathers + traffic-light K / V hoist and, behind the map feature, the destination predictor
    (TB_ENCODE_DEST_SIDE=0 keeps the predictor on the caller's stream: mode "d"); the posterior encoder shares the split functions."""
    import bench
    from trafficbots_amd import synth
    from trafficbots_amd.config import load_model_config

    sd = synth.make_state_dict(7)
    cfg = load_model_config(overrides={"time_step_end": 30, "n_joint_future": 1})
    got = {}
    for mode in ("1", "d", "0"):
        monkeypatch.setenv("TB_ENCODE_SIDE", "0" if mode == "0" else "1")
        monkeypatch.setenv("TB_ENCODE_DEST_SIDE", "0" if mode == "d" else "1")
        c = bench.setup_case(cfg, sd, torch.device("cuda", 0), 0, 5, 48, 80, 1, seed=9900)
        eng, scene = c["eng"], c["scene"]
        encs = [eng.encode_scene(scene) for _ in range(3)]
        torch.cuda.synchronize()
        eng.check_status()
        for e in encs[1:]:
            for k_, v in encs[0].items():
                if torch.is_tensor(v):
                    assert torch.equal(v, e[k_]), (mode, k_)
        got[mode] = {k_: v.clone() for k_, v in encs[0].items() if torch.is_tensor(v)}
    assert set(got["1"]) == set(got["0"]) == set(got["d"]) and len(got["1"]) >= 5
    for k_, v in got["1"].items():
        assert torch.equal(v, got["0"][k_]), k_
        assert torch.equal(v, got["d"][k_]), k_


def test_packed_polyline_tiling_is_bitwise_identical(monkeypatch):
    """The map encoder's polyline block fused (TB_ENCODE_PACK=2, default: two polylines = three row tiles per workgroup, K / V in LDS:
    `k_polyline_fused`) and on the packed tiling (per four polylines four 16-row head tiles + ONE tile with the four
    tail nodes 16 .. 19 of each, instead of eight tiles of which four carry twelve padding rows: tb_encodex_kernels.hip
    `launch_polyline_block_x`) against the padded tiling (TB_ENCODE_PACK=0): every row goes through the same arithmetic, so every
    product of the encoder is bit-identical -- with invalid nodes, wholly invalid polylines and a polyline count that the packed path
    does not take (G % 4 != 0: falls back) and a single scene of 8 polylines."""
    from trafficbots_amd import synth
    from trafficbots_amd.config import load_model_config
    from trafficbots_amd.runtime import HipEngine, scene_from_batch

    sd = synth.make_state_dict(7)
    cfg = load_model_config(overrides={"time_step_end": 30, "n_joint_future": 1})
    eng = HipEngine(cfg)
    eng.load_state_dict(sd)
    for n_scene, n_pl, seed, masks in ((8, 256, 9910, dict(p_invalid_pl=0.3, p_invalid_node=0.4)), (4, 256, 9913, {}),
                                       (5, 260, 9911, dict(p_invalid_pl=0.3, p_invalid_node=0.4)), (3, 65, 9912, dict(p_invalid_node=0.5)), (3, 66, 9915, dict(p_invalid_pl=0.2, p_invalid_node=0.5)), (1, 8, 9914, dict(p_invalid_node=0.5))):
        batch = synth.make_batch(seed, n_scene, n_agent=48, n_pl=n_pl, n_tl=20, **masks)
        scene = scene_from_batch(batch, torch.device("cuda", 0))
        got = {}
        for mode in ("4", "3", "2", "1", "0"):  # ("4", the default: eight waves with merged phases, `k_polyline_fused8<true>`; "3": `<false>`)
            monkeypatch.setenv("TB_ENCODE_PACK", mode)
            enc = eng.encode_scene(scene)
            torch.cuda.synchronize()
            eng.check_status()
            got[mode] = {k_: v.clone() for k_, v in enc.items() if torch.is_tensor(v)}
        assert torch.isfinite(got["2"]["map_feature"]).all()
        if masks.get("p_invalid_pl"):
            assert not bool(got["1"]["map_feature_valid"].all()) and bool(got["1"]["map_feature_valid"].any())
        for k_, v in got["0"].items():
            assert torch.equal(v, got["1"][k_]), (n_pl, k_, "packed")
            assert torch.equal(v, got["2"][k_]), (n_pl, k_, "fused")
            assert torch.equal(v, got["3"][k_]), (n_pl, k_, "fused, eight waves")
            assert torch.equal(v, got["4"][k_]), (n_pl, k_, "fused, eight waves, merged phases")

def test_bench_sub_records_and_traj_err():
    """the default single-GPU command carries the configs[3] / configs[4] sub-records and the golden trajectory error"""
    ln = _run_bench(["--gpus", "1", "--config-steps", "2", "--configs", "k6_bf16", "stress_bf16", "fp32_exact"])
    assert set(ln["configs"]) == {"k6_bf16", "stress_bf16", "fp32_exact"}
    for name, rec in ln["configs"].items():
        assert "error" not in rec, rec
        assert rec["finite"] and rec["value"] > 0 and rec["k_step_fused_us"] > 0
    assert ln["configs"]["k6_bf16"]["instances_per_gpu"] == 192 and ln["configs"]["stress_bf16"]["sim_steps"] == 170
    # round 5: the exact-fp32 kernels' number is part of the driver's line (same workload as the headline, slower kernels)
    ex = ln["configs"]["fp32_exact"]
    assert ex["kernels"]["step"] == "fp32_exact" and ex["encode_ms"] > 0 and ex["value"] < ln["value"] and ex["roofline"]["frac"] > 0
    e = ln["max_abs_traj_err"]
    assert e["flags_equal"] and e["xy_vs_reference_fp32_steps_1_to_60"] <= 1e-4
    assert e["inside_reference_ensemble_every_step"] is True
    # ... and the 8-scene golden, the case at the edge of the rule, is judged in the line as well (VERDICT r04 task 4 (b))
    e8 = e["headline_8"]
    assert e8["flags_equal"] and e8["xy_vs_reference_fp32_steps_1_to_60"] <= 1e-4 and e8["inside_reference_ensemble_every_step"] is True
    assert ln["sustained"] is None or ln["sustained"]["passes"] > 0
    assert ln["ranks"]["ranks_seen"] == 1 and ln["ranks"]["collectives_in_timed_passes"] == 0 and "lib_sha256" in ln
    assert ln["roofline"]["mfma_busy"]["estimated"] > 0 and ln["encode_roofline"]["frac"] > 0


def teardown_module(module):
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "configs_report.json"), "w") as f:
        json.dump(REPORT, f, indent=1)


@pytest.mark.parametrize("seed,case", [
    pytest.param(7, 4, marks=pytest.mark.xfail(strict=True, reason="known-outside fuzz draw (profiles/r05_fuzz_summary.txt: 1.7e-4 m vs the oracle's fp32 "
                                                                     "run at an intermediate step, inside at step 90; the exact-fp32 kernels land in the same place)")),
    pytest.param(7, 13, marks=pytest.mark.xfail(strict=True, reason="known-outside fuzz draw (profiles/r03_fuzz_case13.txt: steps 65-84 outside the oracle ensemble's "
                                                                      "1e-3 prediction limit, rank 0/33, the fp32-MFMA twin likewise)")),
])
def test_known_outside_fuzz_draws(seed, case):
    """VERDICT r05 task 4 (b): the two validation draws of the fuzz campaign (FUZZ_SEED=7, cases 4 and 13) that sit OUTSIDE the
    closed-loop rule against an oracle ensemble measured on the spot -- the same two draws with the same digits in rounds 3, 4 and 5
    -- lived only in a probe's log.  They are promoted here as strict expected failures: they cannot be forgotten, and they turn into
    XPASS (= a red suite, strict) the day a change of the arithmetic moves them inside, which then has to be looked at and recorded.
    Runs the probe itself (tests/probes/gpu_fuzz_validation.py, FUZZ_ONLY: the case is the same case as in a full campaign)."""
    env = dict(os.environ, FUZZ_SEED=str(seed), FUZZ_ONLY=str(case))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "probes", "gpu_fuzz_validation.py"), str(case + 1)], capture_output=True, text=True,
                       env=env, timeout=900, cwd=ROOT)
    tail = "\n".join(r.stdout.strip().splitlines()[-4:])
    REPORT[f"known_outside_fuzz_draw/seed{seed}_case{case}"] = tail
    assert f"case {case:2d}" in r.stdout, r.stdout[-500:] + r.stderr[-500:]   # (the draw ran)
    assert "OUTSIDE" not in r.stdout and r.returncode == 0, tail


def test_default_scheduler_build_runs():
    """VERDICT r05 task 2 (b) / (c): the library built WITHOUT `-amdgpu-sched-strategy=max-ilp` (`__graft_entry__.build_defsched`: same
    sources, the compiler's default machine scheduler) used to die with a GPU memory fault in `k_step_x<false, false>` -- the register
    allocator parked a whole-wave VGPR in an AGPR in front of the exec restore of the divergent `if (tid == 0)` poll of the GRU
    helper's flag (profiles/r06_experiments.txt item 7; tools/isa_waw_lint.py rule 2 flags that placement).  The poll is wave-uniform
    now.  In a SUBPROCESS (a fault would take the process down): two closed-loop goldens against the reference and the guard-page test
    (every caller buffer and every workspace carve between unmapped pages) on that build."""
    lib = os.path.join(ROOT, "trafficbots_amd", "lib", "libtrafficbots_hip_defsched.so")
    if not os.path.exists(lib):
        pytest.skip("no default-scheduler build in this checkout (python -c 'import __graft_entry__ as g; g.build()' makes it)")
    env = dict(os.environ, TB_HIP_LIB=lib)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"), os.path.join(ROOT, "tests", "test_gpu_boundary.py"),
                        "-q", "-m", "gpu", "-p", "no:cacheprovider", "-x",
                        "-k", "(against_reference_golden and (headline_2 or small_k1 or headline_k6)) or no_access_outside"],
                       capture_output=True, text=True, env=env, timeout=1500, cwd=ROOT)
    tail = "\n".join(r.stdout.strip().splitlines()[-6:])
    REPORT["default_scheduler_build"] = tail
    assert r.returncode == 0 and " passed" in tail and "failed" not in tail, tail + r.stderr[-1500:]
