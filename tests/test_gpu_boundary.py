"""GPU tests of the drop-in boundary with the REFERENCE's call forms: Hydra-style construction, the twelve-argument
`encode_input_features`, `model.init`, the stateful `forward(..., state_override=, mask_state_override=)` with the overrides applied
per call, the what-if rollout (`gt_sdc`), and the range guard of the fp16-pair kernels (`tb_check_status`)."""
import copy
import os

import numpy as np
import pytest
import torch

from conftest import ROOT, golden_inputs, load_golden

pytestmark = pytest.mark.gpu


def _small_case(seed=7700, n_scene=3, k=2, step_end=25):
    from trafficbots_amd import synth

    sd = synth.make_state_dict(5)
    scene = dict(n_agent=20, n_pl=50, n_tl=12, p_late_spawn=0.3, p_invalid_agent=0.2, pos_range=140.0)
    batch = synth.make_batch(seed, n_scene, **scene)
    eps = torch.from_numpy(synth.make_latent_noise(5, n_scene * k, 20)).cuda()
    return sd, batch, eps, k, step_end


def test_instantiate_from_a_yaml_shaped_dict_and_reference_call_sequence():
    """(1) `trafficbots_amd.instantiate(cfg)` with a dict shaped like configs/model/traffic_bots.yaml (`_target_` of the reference's
    task module, nested groups, unresolved interpolations) builds the same object as the mirror's own constructor; (2) the
    reference's `test_step` prologue, written the way the reference writes it (`waymo_motion.py:904-921`: input_dict /
    latent_prior_dict comprehension over the pre-processed batch, `encode_input_features(**input_dict)` twice, `pred_goal(...,
    **feature_dict)`, `latent_encoder(**feature_dict)`), gives bitwise the mirror's short form."""
    import trafficbots_amd
    from test_abi_and_host import _yaml_shaped_config
    from trafficbots_amd.waymo_motion import WaymoMotion

    sd, batch, eps, k, step_end = _small_case()
    cfg = _yaml_shaped_config()
    cfg["time_step_end"], cfg["n_joint_future"] = step_end, k
    wm = trafficbots_amd.instantiate(cfg)
    assert isinstance(wm, WaymoMotion) and wm.hparams["model"]["tf_cfg"]["d_model"] == 128
    wm.load_state_dict(sd)
    gen = lambda: torch.Generator(device="cuda").manual_seed(9)  # noqa: E731
    ref = wm.test_step(batch, latent_eps=eps, generator=gen())
    # ---- the reference's own sequence
    b = wm.pre_processing(batch)
    input_dict = {k_.split("input/")[-1]: v for k_, v in b.items() if "input/" in k_}
    latent_prior_dict = {k_.split("latent_prior/")[-1]: v for k_, v in b.items() if "latent_prior/" in k_}
    input_feature_dict = wm.model.encode_input_features(**input_dict)
    latent_prior_feature_dict = wm.model.encode_input_features(**latent_prior_dict)
    assert latent_prior_feature_dict["map_feature"] is input_feature_dict["map_feature"]  # eval-mode aliasing: encoded once
    goal_valid = input_dict["agent_valid"].any(1)
    goal_pred = wm.model.goal_manager.pred_goal(agent_type=b["ref/agent_type"], map_type=b["ref/map_type"],
                                               agent_state=b["ref/agent_state"], **input_feature_dict)
    latent_prior = wm.model.latent_encoder(**latent_prior_feature_dict)
    buf, gs, glp = wm.joint_future_pred(batch=b, input_feature_dict=input_feature_dict, latent=latent_prior, goal=goal_pred,
                                        goal_valid=goal_valid, require_vis_dict=False, latent_eps=eps, generator=gen())
    torch.cuda.synchronize()
    assert torch.equal(buf.preds, ref["rollout_buffer"].preds) and torch.equal(buf.valid, ref["rollout_buffer"].valid)
    assert torch.equal(gs, ref["goal_sample"]) and torch.equal(glp, ref["goal_log_probs"])
    for key in ("map_feature", "agent_feature", "tl_feature"):
        assert torch.equal(input_feature_dict[key], ref["input_feature_dict"][key])


def test_encode_input_features_takes_the_callers_attr_and_pe():
    """tensors the CALLER made in the reference's layout (here: the oracle's restatement of `SceneCentricInput`, torch fp32) go
    through `tb_encode_io.ext_*`: attributes and pose PE are taken as given.  Features agree with the raw-scene path to the
    difference of the two PE evaluations (torch fp32 sin/cos vs the kernel's correctly rounded ones)."""
    from oracle.trafficbots_oracle import Oracle
    from trafficbots_amd import synth
    from trafficbots_amd.waymo_motion import WaymoMotion

    sd = synth.make_state_dict(5)
    batch = synth.make_batch(7800, 2, n_agent=20, n_pl=50, n_tl=12, p_late_spawn=0.3, p_invalid_agent=0.2, p_invalid_node=0.3)
    wm = WaymoMotion(time_step_end=20, n_joint_future=1)
    wm.load_state_dict(sd)
    own = {k: v.clone() for k, v in wm.model.encode_input_features(wm.pre_processing(batch)).items()}
    own_lat, own_logits = wm.model._enc["latent_mean"].clone(), wm.model._enc["dest_logits"].clone()
    inp = Oracle(sd, wm.hparams, torch.float32).preprocess(batch)
    args = {k: inp[k] for k in ("agent_valid", "agent_attr", "agent_pe", "map_valid", "map_attr", "map_pe", "tl_valid", "tl_attr", "tl_pe")}
    args["agent_pos"] = inp["agent_state"][..., :2]
    args["map_pos"] = inp["map_pos"][:, :, 0]
    args["tl_pos"] = torch.from_numpy(batch["history/tl_stop/pos"])
    got = wm.model.encode_input_features(**args)
    torch.cuda.synchronize()
    assert torch.equal(got["map_feature_valid"], own["map_feature_valid"])
    for key in ("map_feature", "agent_feature", "tl_feature"):
        err = (got[key] - own[key]).abs().max().item()
        assert err <= 2e-5, (key, err)
    assert (wm.model._enc["latent_mean"] - own_lat).abs().max().item() <= 2e-5
    fin = torch.isfinite(own_logits)
    assert torch.equal(torch.isfinite(wm.model._enc["dest_logits"]), fin)  # same destination candidates: types read off the attributes
    assert torch.where(fin, wm.model._enc["dest_logits"] - own_logits, torch.zeros_like(own_logits)).abs().max().item() <= 2e-5
    with pytest.raises(ValueError):
        wm.model.encode_input_features(**dict(args, agent_attr=args["agent_attr"][..., :10]))
    with pytest.raises(TypeError):
        wm.model.encode_input_features(args["agent_valid"], args["agent_attr"])


def _open_stepwise(wm, batch, eps, k, step_end, seed=9):
    """the reference's sequence up to the loop: features, `model.init(latent, deterministic)`, simulator opened"""
    from trafficbots_amd.runtime import teacher_forcing_mask

    scene = wm.pre_processing(batch)
    f = wm.model.encode_input_features(scene)
    latent, goal = wm.model.latent_encoder(), wm.model.goal_manager.pred_goal()
    n, a = scene["agent_valid"].shape[0] * k, scene["agent_valid"].shape[2]
    latent.repeat_interleave_(k, 0)
    goal.repeat_interleave_(k, 0)
    det = torch.zeros(n, a, dtype=torch.bool, device="cuda")
    det[::k] = True
    gs = goal.sample(det, generator=torch.Generator(device="cuda").manual_seed(seed))
    feats = dict(scene, map_feature=f["map_feature"], map_feature_valid=f["map_feature_valid"].to(torch.uint8), tl_feature=f["tl_feature"])
    gv = scene["agent_valid"].bool().any(1).repeat_interleave(k, 0)
    mask_tf = teacher_forcing_mask(scene["agent_valid"].bool())
    wm.model.init(latent, det, eps=eps)
    return scene, feats, gs, gv, mask_tf


def test_forward_applies_per_call_state_overrides():
    """the reference's loop, verbatim in structure (`waymo_motion.py:269-306`): per step `state_override = {k: features[k][:, step]}`,
    `mask_state_override = mask_teacher_forcing[:, step]` (zeros past the history), handed to `forward(map_feature=..., ...,
    state_override=..., mask_state_override=...)` -- bitwise the fused rollout; then a DIFFERENT override (a spawn the bound history
    does not contain, at step 14) changes exactly what it must."""
    from trafficbots_amd.waymo_motion import WaymoMotion

    sd, batch, eps, k, step_end = _small_case()
    wm = WaymoMotion(time_step_end=step_end, n_joint_future=k)
    wm.load_state_dict(sd)
    scene, feats, gs, gv, mask_tf = _open_stepwise(wm, batch, eps, k, step_end)
    fused = wm.rollout(feats, None, gs, gv, mask_tf, step_end=step_end, k_futures=k)       # latent / deterministic from model.init
    rep = lambda x: x.repeat_interleave(k, 0)  # noqa: E731  (the reference repeat_interleaves every feature, :523-545)
    features = {"agent_valid": rep(scene["agent_valid"].bool()), "agent_state": rep(scene["agent_state"]),
                "vel": rep(scene["agent_vel"]), "acc": rep(scene["agent_acc"]).unsqueeze(-1), "yaw_rate": rep(scene["agent_yaw_rate"]).unsqueeze(-1),
                "map_feature": rep(feats["map_feature"]), "map_valid": rep(feats["map_feature_valid"].bool()),
                "tl_feature": rep(feats["tl_feature"]), "tl_valid": rep(scene["tl_valid"].bool())}
    mtf = rep(mask_tf)
    state_keys = ("agent_state", "vel", "acc", "yaw_rate")

    def drive(edit=None):
        wm.rollout(feats, None, gs, gv, mask_tf, step_end=step_end, k_futures=k, stepwise=True)
        goal_feature = torch.zeros(mtf.shape[0], mtf.shape[2], 128, device="cuda")  # (bound at open; passed for signature parity)
        for _step in range(1, step_end + 1):
            if _step >= features["agent_valid"].shape[1]:
                mask_state_override = torch.zeros_like(mtf[:, 0])
            else:
                mask_state_override = mtf[:, _step].clone()
            state_override = None
            if mask_state_override.any():
                state_override = {k_: features[k_][:, _step].clone() for k_ in state_keys}
            _gt_valid = None if _step >= features["agent_valid"].shape[1] else features["agent_valid"][:, _step]
            if edit is not None:
                state_override, mask_state_override, _gt_valid = edit(_step, state_override, mask_state_override, _gt_valid)
            step_tl = min(_step - 1, features["tl_valid"].shape[1] - 1)
            state_new, valid_new, train_dict, vis_dict = wm.forward(
                map_feature=features["map_feature"], map_valid=features["map_valid"], tl_feature=features["tl_feature"][:, step_tl],
                tl_valid=features["tl_valid"][:, step_tl], goal_feature=goal_feature, goal_valid=gv,
                state_override=state_override, mask_state_override=mask_state_override, deterministic_action=True,
                require_train_dict=True, require_vis_dict=False, gt_valid=_gt_valid)
            assert state_new.shape == (mtf.shape[0], mtf.shape[2], 4) and valid_new.dtype == torch.bool
            assert train_dict["pred_state"].shape == state_new.shape
        torch.cuda.synchronize()
        return wm.finish_rollout()

    same = drive()
    assert torch.equal(same.preds, fused.preds) and torch.equal(same.valid, fused.valid)
    assert torch.equal(same.override_masks, fused.override_masks)
    for key in ("outside_map", "dest_reached"):
        assert torch.equal(same.violations[key], fused.violations[key])
    # ---- an override the bound history does not contain: instance 1 gets agent 3 re-placed at step 14 (a spawn / teleport)
    tele = torch.tensor([5.0, -7.0, 0.3, 4.0], device="cuda")

    def edit(step, so, m, gtv):
        if step == 14:
            m = m.clone()
            m[1, 3] = True
            if so is None:
                z = torch.zeros(m.shape[0], m.shape[1], 4, device="cuda")
                so = {"agent_state": z.clone(), "vel": z[..., :2].clone(), "acc": z[..., :1].clone(), "yaw_rate": z[..., :1].clone()}
            so["agent_state"][1, 3] = tele
            gtv = torch.ones_like(m) if gtv is None else gtv
        return so, m, gtv

    other = drive(edit)
    assert other.override_masks[1, 3, 13] and not fused.override_masks[1, 3, 13]
    assert torch.equal(other.preds[:, :, :13], fused.preds[:, :, :13])                 # nothing before the step changes
    assert torch.equal(other.preds[0], fused.preds[0])                                 # other instances are untouched
    assert other.valid[1, 3, 14]                                                       # the agent is (re)spawned ...
    d = (other.preds[1, 3, 14, :2] - tele[:2]).norm().item()
    assert d < 1.0, d                                                                  # ... and continues from the forced state
    assert not torch.equal(other.preds[1, :, 20], fused.preds[1, :, 20])               # its neighbours react (interaction attention)
    with pytest.raises(ValueError):
        wm.rollout(feats, None, gs, gv, mask_tf, step_end=step_end, k_futures=k, stepwise=True)
        wm.forward(action_override=torch.zeros(1))  # without its mask
    wm2 = WaymoMotion(time_step_end=step_end, n_joint_future=k)
    wm2.load_state_dict(sd)
    with pytest.raises(RuntimeError):
        wm2.forward()  # no stepwise rollout is open
    # a DEVICE mask with set bits but no state dict: nothing may be forced (the reference would fail on state_override[...] of None)
    wm.rollout(feats, None, gs, gv, mask_tf, step_end=step_end, k_futures=k, stepwise=True)
    st0 = wm.engine.rollout_state()
    wm.forward(state_override=None, mask_state_override=torch.ones_like(mtf[:, 0]).cuda())
    st1 = wm.engine.rollout_state()
    assert torch.equal(st1["agent_valid"].bool() & ~st0["agent_valid"].bool(), torch.zeros_like(st0["agent_valid"]).bool())  # no spawn


@pytest.mark.parametrize("precision", ["fp32", "fp32_exact"])
def test_forward_action_override_against_reference_golden(precision):
    """(`precision="fp32_exact"`: the exact-fp32 step kernel serves the per-call overrides as well since round 5 -- the context a
    range hit falls back to must not refuse a call the default kernels take.)
    `forward(action_override=, mask_action_override=)` (`waymo_motion.py:116-117,174-175` -> `Dynamics.update`, `dynamics.py:96-100`)
    through `tb_rollout_step_ex`: golden `action_override` = the reference's joint_future_pred with its `forward` wrapped to add step s of
    synth.make_action_override to every call (the bound teacher forcing applies unchanged)."""
    from trafficbots_amd import synth
    from trafficbots_amd.runtime import teacher_forcing_mask
    from trafficbots_amd.waymo_motion import WaymoMotion

    g, meta = load_golden("action_override")
    cfg, sd, batch, eps = golden_inputs(meta)
    k, a, step_end = meta["k"], meta["scene"]["n_agent"], meta["time_step_end"]
    n = meta["n_scene"] * k
    n_step = step_end - cfg["time_step_sim_start"] + 1
    ao, am = synth.make_action_override(meta["base_seed"] + 55, n, a, n_step)
    ao, am = torch.from_numpy(ao).cuda(), torch.from_numpy(am).cuda()
    wm = WaymoMotion(time_step_end=step_end, n_joint_future=k, operand_precision=precision)
    wm.load_state_dict(sd)
    assert wm.engine.precision_state()["step"] == ("fp16_pair" if precision == "fp32" else "fp32_exact")
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
    gs = torch.from_numpy(np.transpose(g["goal_sample"], (0, 2, 1)).copy()).reshape(n, a).cuda()
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
        buf = wm.finish_rollout()
        buf.flatten_repeat(k)
        return buf

    buf = drive(True)
    assert (buf.valid.cpu().numpy() == g["valid"]).all() and (buf.override_masks.cpu().numpy() == g["override_masks"]).all()
    for key in ("outside_map", "dest_reached"):
        assert (buf.violations[key].cpu().numpy() == g[key]).all(), key
    assert np.abs(buf.action_log_probs.cpu().numpy() - g["action_log_probs"]).max() <= 2e-5  # (the policy's own action's, unchanged)
    d = np.abs(buf.preds.cpu().numpy() - g["preds"]) * g["valid"][..., None]
    assert d[..., :10, :2].max() <= 1e-5
    # the one closed-loop rule (tools/ensemble.py::closed_loop_rule) against the golden's independent reference ensemble
    from tools import ensemble

    e = np.load(os.path.join(ROOT, "tests", "golden", "ensg", "action_override.npz"))
    both = (g["valid"] & g["valid_fp64"])[..., None]
    d64 = (np.abs(buf.preds.cpu().numpy().astype(np.float64) - g["preds_fp64"]) * both)[..., :2].max(axis=(0, 1, 2, 4))
    per_step = d[..., :2].max(axis=(0, 1, 2, 4))
    r = ensemble.closed_loop_rule(per_step, d64, e["ensg_d32"], e["ensg_d64"], n_flat=per_step.shape[0])
    assert r["ok"], {k_: v for k_, v in r.items() if k_ != "per_step"}
    plain = drive(False)
    assert float((plain.preds - buf.preds).abs().max()) > 0.05  # the override really steers


def test_device_samplers_against_torch_distributions():
    """SURVEY 8 rows a9 / a10 on the device (`tb_latent_sample`, `tb_dest_sample`, csrc/tb_sample_kernels.hip) against
    torch.distributions on the host -- what the reference's `DiagGaussian` / `DestCategorical` wrap (`distributions.py:40-59,158-201`)."""
    from torch.distributions import Categorical, Independent, Normal

    from trafficbots_amd import synth
    from trafficbots_amd.distributions import DestCategorical, DiagGaussian
    from trafficbots_amd.waymo_motion import WaymoMotion

    wm = WaymoMotion(time_step_end=12, n_joint_future=1)
    wm.load_state_dict(synth.make_state_dict(7))
    gen = torch.Generator().manual_seed(1)
    b, k, a, p = 3, 4, 21, 70
    mean, log_std = torch.randn(b, a, 16, generator=gen), torch.linspace(-2.0, 0.5, 16)
    eps = torch.randn(b * k, a, 16, generator=gen)
    det = torch.rand(b * k, a, generator=gen) < 0.4
    d = DiagGaussian(mean.cuda(), log_std.cuda(), engine=wm.engine)
    d.repeat_interleave_(k, 0)
    z = d.sample(det.cuda(), eps=eps.cuda()).cpu()
    mean_n = mean.repeat_interleave(k, 0)
    want = torch.where(det[..., None], mean_n, mean_n + eps * log_std.exp())
    assert torch.equal(z[det], mean_n[det]) and (z - want).abs().max() <= 1e-6
    assert torch.equal(d.sample(True).cpu(), mean_n)                       # bool forms: everybody the mean / everybody drawn
    assert (d.sample(False, eps=eps.cuda()).cpu() - (mean_n + eps * log_std.exp())).abs().max() <= 1e-6
    lp = d.log_prob(z.cuda()).cpu()
    assert (lp - Independent(Normal(mean_n, log_std.exp()), 1).log_prob(z)).abs().max() <= 2e-4  # (values reach ~ -60: relative 3e-6)
    # the parameter form: log_std NULL -> the loaded prior / posterior vectors
    zp, lpp = wm.engine.latent_sample(mean.cuda(), k, eps=eps.cuda(), deterministic=det.cuda(), posterior=True)
    ls_post = torch.from_numpy(synth.make_state_dict(7)["model.latent_encoder.latent_post_dist.log_std"])
    assert (zp.cpu() - torch.where(det[..., None], mean_n, mean_n + eps * ls_post.exp())).abs().max() <= 1e-6
    assert (lpp.cpu() - Independent(Normal(mean_n, ls_post.exp()), 1).log_prob(zp.cpu())).abs().max() <= 2e-4

    # ---- destinations: masked logits as DestPredictor.forward leaves them (-inf for inadmissible polylines, zeros for invalid agents)
    logits = 3.0 * torch.randn(b, a, p, generator=gen)
    logits[torch.rand(b, a, p, generator=gen) < 0.5] = float("-inf")
    logits[0, 0] = 0.0
    logits[1, 2, 1:] = float("-inf")                                       # a single admissible polyline
    logits[(logits == float("-inf")).all(-1)] = 0.0
    c = DestCategorical(logits.cuda(), engine=wm.engine)
    ref = Categorical(logits=logits)
    assert torch.equal(c.sample(True).cpu(), ref.probs.argmax(-1))
    assert (c.probs.cpu() - ref.probs).abs().max() <= 1e-6
    s0 = c.sample(True)
    assert (c.log_prob(s0).cpu() - ref.log_prob(s0.cpu())).abs().max() <= 2e-6          # Categorical(logits=): logits - logsumexp
    c.repeat_interleave_(k, 0)
    rep = Categorical(probs=ref.probs.repeat_interleave(k, 0))                           # what the reference holds after the repeat
    u = torch.rand(b * k, a, generator=gen)
    s = c.sample(det.cuda(), u=u.cuda()).cpu()
    assert torch.equal(s[det], rep.probs.argmax(-1)[det])
    pr = ref.probs.repeat_interleave(k, 0).double()
    cdf = pr.cumsum(-1)
    inv = (cdf > (u.double() * cdf[..., -1])[..., None]).float().argmax(-1)             # smallest j with cdf_j > u * total
    differ = (s != inv) & ~det
    if differ.any():  # only where u sits on a CDF step within fp32 rounding
        gap = (cdf.gather(-1, torch.minimum(s, inv)[..., None])[..., 0] - u.double() * cdf[..., -1]).abs()
        assert (gap[differ] <= 1e-5).all(), float(gap[differ].max())
    assert (pr.gather(-1, s[..., None]) > 0).all(), "a masked polyline was sampled"
    assert (c.log_prob(s.cuda()).cpu() - rep.log_prob(s)).abs().max() <= 2e-6
    masked = torch.zeros(b * k, a, dtype=torch.long)                                    # index 0 is masked for many rows
    lpm = c.log_prob(masked.cuda()).cpu()
    assert (lpm - rep.log_prob(masked)).abs().max() <= 2e-6                              # clamped at log(eps), never -inf
    # frequencies of many draws follow the probabilities (one row, 20000 uniforms through the K axis)
    row = torch.tensor([[[0.0, 1.0, float("-inf"), 2.0, -1.0, 0.5]]])
    many = DestCategorical(row.cuda(), engine=wm.engine)
    many.repeat_interleave_(20000, 0)
    draws = many.sample(False, generator=torch.Generator(device="cuda").manual_seed(5)).cpu().flatten()
    freq = torch.bincount(draws, minlength=6).double() / draws.numel()
    assert freq[2] == 0 and (freq - torch.softmax(row[0, 0], -1).double()).abs().max() <= 0.015


def test_require_vis_dict_against_reference_golden():
    """`joint_future_pred(..., require_vis_dict=True)` -> `rollout` -> `forward(require_vis_dict=True)` (`waymo_motion.py:167,191-201,305`):
    per step the applied action, the navigator's goal validity and the head-mean attention weights of the three blocks, served by the
    un-fused `tb_forward` next to the fused step launches.  Golden `vis_dict` = the reference's own buffer (`RolloutBuffer.vis_dicts`
    after `finish` + `flatten_repeat`); the trajectories of the per-step-driven rollout equal the fused rollout's bit for bit."""
    from trafficbots_amd.waymo_motion import WaymoMotion

    g, meta = load_golden("vis_dict")
    cfg, sd, batch, eps = golden_inputs(meta)
    wm = WaymoMotion(time_step_end=meta["time_step_end"], n_joint_future=meta["k"])
    wm.load_state_dict(sd)
    gs = torch.from_numpy(np.transpose(g["goal_sample"], (0, 2, 1)).copy())
    eps_t = torch.from_numpy(eps).cuda()

    def run(vis):
        scene = wm.pre_processing(batch)
        scene.pop("gt", None)
        feats = wm.model.encode_input_features(scene)
        buf, _, _ = wm.joint_future_pred(scene, feats, wm.model.latent_encoder(), wm.model.goal_manager.pred_goal(),
                                         scene["agent_valid"].bool().any(1), require_vis_dict=vis, latent_eps=eps_t, goal_sample=gs)
        torch.cuda.synchronize()
        return buf

    buf = run(True)
    fused = run(False)
    assert torch.equal(buf.preds, fused.preds) and torch.equal(buf.valid, fused.valid) and not fused.vis_dicts
    assert (buf.valid.cpu().numpy() == g["valid"]).all()
    assert np.abs(buf.preds.cpu().numpy() - g["preds"]).max() <= 2e-5
    v = g["valid"]  # [B,A,K,S]: validity BEFORE each step = the rows the reference computes weights for
    assert set(buf.vis_dicts) == {"action", "goal_valid", "attn_weights_to_pl", "attn_weights_to_tl", "attn_weights_to_agent"}
    for key, tol in (("action", 2e-5), ("attn_weights_to_pl", 5e-6), ("attn_weights_to_tl", 5e-6), ("attn_weights_to_agent", 5e-6)):
        got, ref = buf.vis_dicts[key].numpy(), g["vis/" + key]
        assert got.shape == ref.shape, (key, got.shape, ref.shape)
        assert (np.abs(got - ref) * v[..., None]).max() <= tol, (key, float((np.abs(got - ref) * v[..., None]).max()))
    assert (buf.vis_dicts["goal_valid"].numpy() == g["vis/goal_valid"]).all()
    w = buf.vis_dicts["attn_weights_to_pl"].numpy()
    assert (np.abs(w.sum(-1) - np.round(w.sum(-1))) * v).max() <= 1e-5  # rows sum to 1 (or 0: no admissible key)
    # the reference's call form: forward(require_vis_dict=True) on a simulator that was not opened for it raises
    scene = wm.pre_processing(batch)
    scene.pop("gt", None)
    wm.model.encode_input_features(scene)
    with pytest.raises(NotImplementedError):
        wm.rollout(dict(scene), None, gs.reshape(-1, gs.shape[-1]).cuda(), scene["agent_valid"].bool().any(1).repeat_interleave(meta["k"], 0),
                   scene["agent_valid"].bool(), require_vis_dict=True, stepwise=True, k_futures=meta["k"])


def test_what_if_rollout_forces_the_sdc_trajectory():
    """`rollout(..., gt_sdc=...)` (`waymo_motion.py:279-284`): agent 0 is teacher-forced to the given trajectory at EVERY step, the
    other agents run closed loop around it."""
    from trafficbots_amd.waymo_motion import WaymoMotion

    sd, batch, eps, k, step_end = _small_case(k=1)
    batch["history/agent/valid"][:, :, 0] = True  # the SDC exists
    wm = WaymoMotion(time_step_end=step_end, n_joint_future=k)
    wm.load_state_dict(sd)
    scene, feats, gs, gv, mask_tf = _open_stepwise(wm, batch, eps, k, step_end)
    base = wm.rollout(feats, None, gs, gv, mask_tf, step_end=step_end, k_futures=k)
    n = scene["agent_valid"].shape[0]
    t = torch.arange(step_end + 1, device="cuda", dtype=torch.float32)
    sdc_state = torch.stack([10.0 + 0.8 * t, -20.0 + 0.1 * t, torch.full_like(t, 0.12), torch.full_like(t, 8.0)], -1).expand(n, -1, -1).contiguous()
    gt_sdc = {"agent_state": sdc_state, "vel": torch.zeros(n, step_end + 1, 2, device="cuda") + 0.5,
              "acc": torch.zeros(n, step_end + 1, 1, device="cuda"), "yaw_rate": torch.zeros(n, step_end + 1, 1, device="cuda")}
    wi = wm.rollout(feats, None, gs, gv, mask_tf, step_end=step_end, k_futures=k, gt_sdc=gt_sdc)
    torch.cuda.synchronize()
    assert wi.override_masks[:, 0].all()
    assert torch.equal(wi.final["final_state"][:, 0], sdc_state[:, step_end])
    # the prediction for step t+1 starts from the forced state of step t: within one step's motion of the given trajectory
    gap = (wi.preds[:, 0, 1:, :2] - sdc_state[:, 2:, :2]).norm(dim=-1)
    assert gap.max() < 1.5, gap.max()
    assert not torch.equal(wi.preds[:, 1:], base.preds[:, 1:])  # the others see a different SDC


def _scaled(sd, s):
    """FFN hidden activations of the three as2pl layers scaled by `s` with the mathematically identical network: W1, b1 *= s,
    W2 /= s (ReLU is positively homogeneous)"""
    sd = copy.deepcopy(sd)
    for i in range(3):
        p = f"model.transformer_as2pl.layers.{i}."
        sd[p + "linear1.weight"] = sd[p + "linear1.weight"] * np.float32(s)
        sd[p + "linear1.bias"] = sd[p + "linear1.bias"] * np.float32(s)
        sd[p + "linear2.weight"] = sd[p + "linear2.weight"] / np.float32(s)
    return sd


@pytest.mark.parametrize("scale,expect", [(3e3, "ok"), (3e-3, "ok"), (1e-4, "small"), (3e5, "flag"), (1e-6, "weights")])
def test_fp16_pair_operand_range(scale, expect):
    """The fp32-accurate kernels carry GEMM operands as fp16 pairs: full accuracy (2^-22 relative) for magnitudes in
    ~[1.2e-4, 65504), an absolute floor of 2.9e-11 per operand below.  The as2pl FFN hidden layer is scaled (same function,
    `_scaled`) so that its activations -- GEMM operands -- reach ~1e3..1e4, ~1e-3, ~1e-4 and ~1e5+, and the one-step policy feature /
    a 15-step rollout are compared with the fp64 oracle of the SAME scaled weights:
      ok      : as accurate as the unscaled network (<= 1e-5 on the re-synced policy feature), no flag, fp16-pair kernels;
      small   : hidden activations at the edge of fp16's normal range (the compensating W2 is ~1e3): error reported, <= 1e-4;
      flag    : beyond 65504 the step is RE-RUN on the exact-fp32 kernels (round 4; `tb_check_status` returns 3, the context switches
                itself, `WaymoMotion` re-issues the step and warns with the measured slowdown): the result is as accurate as `ok`;
      weights : a WEIGHT beyond the range (here W2 / 1e-6 ~ 1e5) selects the exact-fp32 kernels at `tb_finalize_weights` (stderr note,
                `tb_precision_state`): same accuracy.  The reference has no range limit (`src/models/modules/mlp.py:20-85`)."""
    from oracle.trafficbots_oracle import Oracle
    from trafficbots_amd import synth
    from trafficbots_amd.waymo_motion import WaymoMotion

    g, meta = load_golden("small_k1")
    cfg, sd0, batch, eps = golden_inputs(meta)
    step_end = 15
    sd = _scaled(sd0, scale)
    wm = WaymoMotion(time_step_end=step_end, n_joint_future=1)
    wm.load_state_dict(sd)
    st = wm.engine.precision_state()
    if expect == "weights":
        assert st["step"] == "fp32_exact" and st["encode"] == "fp32_exact" and st["weight_out_of_range"] and "linear2.weight" in st["note"]
        wb = WaymoMotion(time_step_end=step_end, n_joint_future=1, operand_precision="bf16")  # (step kernels in bf16: fp32's range ...
        wb.load_state_dict(sd)                                                                 # ... the encoders leave their fp16 pairs)
        sb = wb.engine.precision_state()
        assert sb["step"] == "bf16" and sb["encode"] == "fp32_exact" and sb["weight_out_of_range"]
        assert torch.isfinite(wb.test_step(batch, latent_eps=torch.from_numpy(eps).cuda())["rollout_buffer"].preds).all()
    else:
        assert st["step"] == "fp16_pair" and st["encode"] == "fp16_pair" and not st["weight_out_of_range"]
    if expect == "flag":
        # the same activations on a bf16 context: its step kernels have fp32's range, its scene encoders keep fp16 pairs, share the
        # as2pl weights (latent encoder) and DO overflow -- the encoders alone fall back, the message names the stage
        wb = WaymoMotion(time_step_end=step_end, n_joint_future=1, operand_precision="bf16")
        wb.load_state_dict(sd)
        wb.fallback_policy = wm.fallback_policy = "sticky"  # (what follows is about the switch itself; the default policy: below)
        with pytest.warns(RuntimeWarning, match="scene encoders"):
            o = wb.test_step(batch, latent_eps=torch.from_numpy(eps).cuda())
        assert torch.isfinite(o["rollout_buffer"].preds).all()
        sb = wb.engine.precision_state()
        assert sb["step"] == "bf16" and sb["encode"] == "fp32_exact" and sb["activation_overflow"]
        # without the harness (check_range off) the overflow is the caller's to handle: the check raises and the context has switched
        wr = WaymoMotion(time_step_end=step_end, n_joint_future=1)
        wr.load_state_dict(sd)
        wr.check_range = False
        wr.test_step(batch, latent_eps=torch.from_numpy(eps).cuda())
        with pytest.raises(RuntimeError, match="65504"):
            wr.engine.check_status()
        assert wr.engine.precision_state()["step"] == "fp32_exact"
        wr.engine.check_status()  # the flag is cleared by the check that reported it
        # round 5: a step that draws from a `generator` and is re-run sees the SAME draws (the generator's state
        # is restored for the re-run) -- the result equals, bit for bit, what the same context (now on the kernels it switched to)
        # computes from a fresh generator with that seed (K = 2: the second future's personality and destination are sampled)
        wg = WaymoMotion(time_step_end=step_end, n_joint_future=2)
        wg.load_state_dict(sd)
        wg.fallback_policy = "sticky"
        with pytest.warns(RuntimeWarning, match="re-run on the exact-fp32 kernels"):
            og = wg.test_step(batch, generator=torch.Generator(device="cuda").manual_seed(77))
        og = {"preds": og["rollout_buffer"].preds.clone(), "goal_sample": og["goal_sample"].clone()}
        import warnings as _w

        with _w.catch_warnings():
            _w.simplefilter("error", RuntimeWarning)  # (no second fallback)
            o2 = wg.test_step(batch, generator=torch.Generator(device="cuda").manual_seed(77))
        assert torch.equal(og["goal_sample"], o2["goal_sample"]) and torch.equal(og["preds"], o2["rollout_buffer"].preds)
        o3 = wg.test_step(batch, generator=torch.Generator(device="cuda").manual_seed(78))
        assert not torch.equal(og["preds"], o3["rollout_buffer"].preds)  # (the draws matter)
        with pytest.warns(RuntimeWarning, match=r"re-run on the exact-fp32 kernels: .* ms against"):
            out = wm.test_step(batch, latent_eps=torch.from_numpy(eps).cuda(), tap_step=1)
        st = wm.engine.precision_state()
        assert st["step"] == "fp32_exact" and st["activation_overflow"]
        # round 6, the DEFAULT policy ("adaptive") with weights that overflow on every batch: the first two re-runs go back to the fast
        # kernels, the third overflow of the window keeps the context on the exact ones -- no fourth re-run; every result is the exact
        # kernels' (bit-identical to the sticky context's)
        wa = WaymoMotion(time_step_end=step_end, n_joint_future=1)
        wa.load_state_dict(sd)
        assert wa.fallback_policy == "adaptive" and wa.fallback_sticky_after == 3
        e_ = torch.from_numpy(eps).cuda()
        for i in range(2):
            with pytest.warns(RuntimeWarning, match="back on the fp16-pair kernels"):
                oa = wa.test_step(batch, latent_eps=e_, tap_step=1)
            assert torch.equal(oa["rollout_buffer"].preds, out["rollout_buffer"].preds)
            sa = wa.engine.precision_state()
            assert sa["step"] == "fp16_pair" and sa["encode"] == "fp16_pair" and not sa["activation_overflow"] and sa["note"] == ""
        with pytest.warns(RuntimeWarning, match=r"stays on them \(3 of the last 3 checked steps overflowed\)"):
            oa = wa.test_step(batch, latent_eps=e_, tap_step=1)
        assert wa.engine.precision_state()["step"] == "fp32_exact" and wa.n_fallbacks == 3
        with _w.catch_warnings():
            _w.simplefilter("error", RuntimeWarning)
            oa = wa.test_step(batch, latent_eps=e_, tap_step=1)
        assert torch.equal(oa["rollout_buffer"].preds, out["rollout_buffer"].preds)
    else:
        import warnings

        with warnings.catch_warnings():
            warnings.simplefilter("error", RuntimeWarning)  # (no fallback in the other regimes)
            out = wm.test_step(batch, latent_eps=torch.from_numpy(eps).cuda(), tap_step=1)
    torch.cuda.synchronize()
    buf = out["rollout_buffer"]
    dest = out["goal_sample"].transpose(1, 2).reshape(meta["n_scene"], -1).cpu().numpy()
    cfg15 = dict(cfg, time_step_end=step_end)
    with torch.no_grad():
        r64 = Oracle(sd, cfg15, torch.float64, hoist=True).joint_future_pred(batch, 1, eps, step_end, dest_override=dest, tap_steps=(1,))
    err_f = (buf.taps["tap_policy_feature"].cpu().double() - r64["tap1/policy_feature"]).abs().max().item()
    err_xy = ((buf.preds.cpu().double() - r64["preds"]).abs() * r64["valid"].unsqueeze(-1))[..., :2].max().item()
    assert torch.isfinite(buf.preds).all()
    assert (buf.valid.cpu() == r64["valid"]).all()
    import json
    import os

    from conftest import ROOT

    path = os.path.join(ROOT, "gpurun_out", "range_report.json")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    rep = json.load(open(path)) if os.path.exists(path) else {}
    rep[f"scale_{scale:g}"] = {"policy_feature_step1_vs_fp64": err_f, "xy_15_steps_vs_fp64": err_xy}
    json.dump(rep, open(path, "w"), indent=1)
    if expect in ("ok", "flag", "weights"):  # (flag / weights: the exact-fp32 kernels produced this result)
        assert err_f <= 1e-5 and err_xy <= 1e-4, (expect, err_f, err_xy)
    else:
        assert err_f <= 1e-4 and err_xy <= 1e-3, (err_f, err_xy)


def test_a_downgraded_context_gets_its_kernels_back_with_in_range_weights():
    """the switch to the exact-fp32 kernels was sticky -- a later `tb_finalize_weights` with in-range weights never
    restored the XDL kernels.  Every load now starts from the configured selection; and a switch that happens while a stepwise
    rollout is open closes that rollout (its workspace holds XDL-ordered K / V the exact kernel cannot read)."""
    from trafficbots_amd import synth
    from trafficbots_amd.waymo_motion import WaymoMotion

    sd = synth.make_state_dict(5)
    big = {k: (v * 1e7 if k.endswith("transformer_as2pl.layers.0.linear2.weight") else v) for k, v in sd.items()}
    wm = WaymoMotion(time_step_end=20, n_joint_future=1)
    wm.load_state_dict(big)
    st = wm.engine.precision_state()
    assert st["step"] == "fp32_exact" and st["weight_out_of_range"]
    wm.load_state_dict(sd)
    st = wm.engine.precision_state()
    assert st["step"] == "fp16_pair" and st["encode"] == "fp16_pair" and not st["weight_out_of_range"], st


@pytest.mark.parametrize("case", ["masks", "degenerate"])
def test_traffic_bots_forward_with_attention_weights_against_reference_golden(case):
    """`TrafficBots.forward(agent_valid, agent_feature, map_valid, map_feature, tl_valid, tl_feature, goal_valid, goal_feature,
    need_weights=True) -> (policy_feature, latent_logp, attn_pl, attn_tl, attn_agent)` (`traffic_bots.py:163-247`) through `tb_forward`
    against tests/golden/forward_weights.npz -- the reference's own calls inside a rollout: inputs, recurrent state, all five outputs."""
    import json
    import os

    from conftest import GOLDEN_DIR
    from trafficbots_amd import synth
    from trafficbots_amd.distributions import DiagGaussian
    from trafficbots_amd.waymo_motion import WaymoMotion

    g = np.load(os.path.join(GOLDEN_DIR, "forward_weights.npz"))
    meta = json.loads(bytes(g[f"{case}/meta_json"]).decode())
    wm = WaymoMotion(time_step_end=meta["time_step_end"], n_joint_future=meta["k"])
    wm.load_state_dict(synth.make_state_dict(meta["weight_seed"]))
    t = lambda k: torch.from_numpy(g[k]).cuda()  # noqa: E731
    for s_ in meta["steps"]:
        p = f"{case}/step{s_}/"
        out = wm.engine.forward_trunk(t(p + "agent_valid"), t(p + "agent_feature"), t(p + "map_valid"), t(p + "map_feature"), t(p + "tl_valid"),
                                      t(p + "tl_feature"), t(p + "goal_valid"), t(p + "goal_feature"), t(p + "latent_sample"),
                                      None if bool(g[p + "hidden_in_is_none"]) else t(p + "hidden_in"), need_weights=True)
        torch.cuda.synchronize()
        v = g[p + "agent_valid"][..., None]
        assert np.abs(out["policy_feature"].cpu().numpy() - g[p + "policy_feature"]).max() <= 2e-5
        assert np.abs(out["hidden"].cpu().numpy() - g[p + "hidden_out"]).max() <= 2e-5
        for key in ("attn_pl", "attn_tl", "attn_agent"):
            got = out[key].cpu().numpy()
            assert (np.abs(got - g[p + key]) * v).max() <= 2e-6, (s_, key)
            assert (np.abs(got.sum(-1) - np.round(got.sum(-1))) * v[..., 0]).max() <= 1e-5  # rows sum to 1 (or 0: no admissible key / bypass)
    # the mirror class, stateful like the reference: init -> forward (samples the personality, hidden None) -> forward (hidden carried)
    p = f"{case}/step{meta['steps'][0]}/"
    n, a = g[p + "agent_valid"].shape
    lat = DiagGaussian(t(p + "latent_sample"), torch.full((16,), -1.0, device="cuda"), valid=t(p + "agent_valid"))
    wm.model.init(lat, True)
    args = [t(p + k) for k in ("agent_valid", "agent_feature", "map_valid", "map_feature", "tl_valid", "tl_feature", "goal_valid", "goal_feature")]
    pf, logp, w_pl, w_tl, w_ag = wm.model(*args, need_weights=True)
    assert logp.shape == (n, a) and w_pl.shape == (n, a, g[p + "map_valid"].shape[1]) and w_ag.shape == (n, a, a)
    if bool(g[p + "hidden_in_is_none"]):
        assert np.abs(pf.cpu().numpy() - g[p + "policy_feature"]).max() <= 2e-5
    pf2, _, none_pl, _, _ = wm.model(*args)  # second call: the hidden state moved on, no weights asked for
    assert none_pl is None and float((pf2 - pf).abs().max()) > 1e-4


def test_unfused_forward_cross_checks_the_fused_step_kernel():
    """Two independent device implementations of the same trunk: the policy feature the fused step kernel taps at step 1 (fp16-pair
    MFMA, hoisted K / V, fused epilogues) against `tb_forward` (plain fp32 FMA kernels on the reference's row-major tensors) fed with
    the fused kernel's own agent feature of that step."""
    from trafficbots_amd import synth
    from trafficbots_amd.waymo_motion import WaymoMotion

    k = 2
    sd = synth.make_state_dict(6)
    batch = synth.make_batch(9900, 3, n_agent=20, n_pl=50, n_tl=12, p_invalid_agent=0.2, p_late_spawn=0.2, pos_range=120.0)
    eps = torch.from_numpy(synth.make_latent_noise(9901, 3 * k, 20)).cuda()
    wm = WaymoMotion(time_step_end=12, n_joint_future=k)
    wm.load_state_dict(sd)
    out = wm.test_step(batch, latent_eps=eps, generator=torch.Generator(device="cuda").manual_seed(1), tap_step=1)
    buf, feats = out["rollout_buffer"], out["input_feature_dict"]
    scene = wm.pre_processing(batch)
    rep = lambda x: x.repeat_interleave(k, 0)  # noqa: E731
    agent_valid = rep(scene["agent_valid"][:, 0].bool())                      # validity going into step 1 = history step 0
    dest = out["goal_sample"].transpose(1, 2).reshape(3 * k, -1)             # [N,A]
    map_f = rep(feats["map_feature"])
    goal_feature = torch.gather(map_f, 1, dest.long().unsqueeze(-1).expand(-1, -1, 128))
    goal_valid = rep(scene["agent_valid"].bool().any(1))
    fw = wm.engine.forward_trunk(agent_valid, buf.taps["tap_agent_feature"], rep(feats["map_feature_valid"]), map_f,
                                 rep(scene["tl_valid"][:, 0].bool()), rep(feats["tl_feature"][:, 0]), goal_valid, goal_feature,
                                 buf.latent_sample, None)
    torch.cuda.synchronize()
    d = (fw["policy_feature"] - buf.taps["tap_policy_feature"]).abs() * agent_valid.unsqueeze(-1)
    assert float(d.max()) <= 5e-6 * max(1.0, float(buf.taps["tap_policy_feature"].abs().max())), float(d.max())


def test_range_flag_with_two_contexts_on_one_device():
    """the fp16-pair range flag is one word per device and kernel family.  With two contexts on one device, the check of
    context A used to take -- and clear -- an overflow raised by context B: A was downgraded for nothing and B got rc 0 for invalid
    results.  Round 5: whoever takes the flag hands its bits to every other live context of the device; B's own check still reports
    the overflow and switches B to the exact kernels (A's report is conservative: it cannot tell whose launch raised the flag)."""
    from trafficbots_amd.waymo_motion import WaymoMotion

    g, meta = load_golden("small_k1")
    cfg, sd0, batch, eps = golden_inputs(meta)
    wa = WaymoMotion(time_step_end=15, n_joint_future=1)
    wa.load_state_dict(sd0)
    wb = WaymoMotion(time_step_end=15, n_joint_future=1)
    wb.load_state_dict(_scaled(sd0, 3e5))
    wa.check_range = wb.check_range = False
    e = torch.from_numpy(eps).cuda()
    wa.test_step(batch, latent_eps=e)
    wa.engine.check_status()  # clean
    wb.test_step(batch, latent_eps=e)  # activations of ~1e5: raises the device's flag
    with pytest.raises(RuntimeError, match="65504"):
        wa.engine.check_status()  # A's check takes the flag ...
    with pytest.raises(RuntimeError, match="65504"):
        wb.engine.check_status()  # ... and B still hears of it
    assert wb.engine.precision_state()["step"] == "fp32_exact" and wb.engine.precision_state()["activation_overflow"]
    wa.engine.check_status()
    wb.engine.check_status()  # both clean now
    out = wb.test_step(batch, latent_eps=e)
    wb.engine.check_status()
    assert torch.isfinite(out["rollout_buffer"].preds).all()


@pytest.mark.parametrize("at_end", [0, 1])
def test_no_access_outside_the_callers_buffers(at_end):
    """Round 5 (profiles/r05_experiments.txt item 16: a kernel read 8 bytes in FRONT of `preds` -- silent until an allocation boundary
    sat there).  tests/probes/gpu_guard_pages.py runs test_step (K = 1 .. 3, sampled actions), validation_step and a stepwise rollout
    over three shapes with EVERY tensor handed to the C ABI replaced by a copy that has unmapped address space on both sides
    (tests/guard/tb_guard.cpp: hipMemAddressReserve / hipMemMap; 4 KiB granules), at the start of its mapping (at_end = 0: an access in
    front of an array is a GPU memory fault) or at its end (at_end = 1: an access behind it is).  A fault aborts the subprocess.  The
    build with the bug put back (`tools/build_variant.sh _oobbug -DTB_DBG_OOB_STEP_INDEX`) dies here with "Memory access fault"."""
    import subprocess
    import sys

    from conftest import ROOT

    import __graft_entry__ as g

    g.build_guard()
    # ... and every carve of the library's OWN workspace in a mapping of its own between unmapped pages (TB_WS_GUARD: 1 = the buffer ends
    # its mapping, 2 = it opens it): an overrun from one internal buffer into the next cannot hide inside the one allocation
    env = dict(os.environ, GUARD_AT_END=str(at_end), TB_WS_GUARD="1" if at_end else "2")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "probes", "gpu_guard_pages.py")], env=env, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0 and "GUARD-OK" in r.stdout, (r.returncode, r.stdout[-400:], r.stderr[-1200:])
    assert "TB_WS_GUARD=" in r.stderr and "every workspace carve" in r.stderr  # (the library's debug mode was on)



def _engine(cfg_overrides, sd):
    from trafficbots_amd.waymo_motion import WaymoMotion

    wm = WaymoMotion(**cfg_overrides)
    wm.load_state_dict(sd)
    return wm


def _dev_batch(batch):
    return {k: torch.from_numpy(np.asarray(v)).cuda() for k, v in batch.items()}


def _same_step_outputs(a, b, what):
    ra, rb = a["rollout_buffer"], b["rollout_buffer"]
    for name in ("preds", "valid", "override_masks", "action_log_probs", "latent_log_probs"):
        assert torch.equal(getattr(ra, name), getattr(rb, name)), (what, name)
    for k_ in ra.violations:
        assert torch.equal(ra.violations[k_], rb.violations[k_]), (what, k_)
    for name in ("goal_sample", "goal_log_probs", "scores", "latent_mean", "dest_logits"):
        assert torch.equal(a[name], b[name]), (what, name)
    for k_, v in a["pred_dict"].items():
        if torch.is_tensor(v):
            assert torch.equal(v, b["pred_dict"][k_]), (what, k_)


def test_staged_and_prefetched_steps_equal_the_device_side_conversion():
    """VERDICT r05 task 1: the drop-in `test_step(batch)` with the batch STAGED on the host (one pinned slab, one upload, no device-side
    conversion kernels: staging.py), and with the next batch staged + encoded on a side stream under the current rollout
    (`wm.prefetch`), against the torch-op conversion of a batch that already lives on the device (the path of rounds 1-5): every
    output bit-identical, for K = 3 futures with drawn destinations, over a stream of distinct batches; `validation_step` likewise
    (ground truth staged with the scene).  One upload per staged batch."""
    from trafficbots_amd import synth

    sd = synth.make_state_dict(5)
    scene = dict(n_agent=20, n_pl=50, n_tl=12, p_late_spawn=0.3, p_invalid_agent=0.2, pos_range=140.0)
    batches = [synth.make_batch(7800 + i, 3, **scene) for i in range(5)]
    k, step_end = 3, 30
    eps = torch.from_numpy(synth.make_latent_noise(5, 3 * k, 20)).cuda()
    wm = _engine({"time_step_end": step_end, "n_joint_future": k}, sd)
    gen = lambda i: torch.Generator(device="cuda").manual_seed(100 + i)  # noqa: E731
    ref = [wm.test_step(_dev_batch(b), latent_eps=eps, generator=gen(i)) for i, b in enumerate(batches)]
    stager = wm.engine.stager(wm._tf_params)
    n0 = stager.n_uploads
    staged = [wm.test_step(b, latent_eps=eps, generator=gen(i)) for i, b in enumerate(batches)]
    assert stager.n_uploads - n0 == len(batches)
    pf = wm.prefetch(batches)
    fetched = [wm.test_step(sb, latent_eps=eps, generator=gen(i)) for i, sb in enumerate(pf)]
    assert pf.n_staged == len(batches) == len(fetched)
    wm.check_range = False  # (no synchronisation at the end of a step: the host runs ahead, the prefetcher's events order the streams)
    fetched2 = [wm.test_step(sb, latent_eps=eps, generator=gen(i)) for i, sb in enumerate(wm.prefetch(batches))]
    torch.cuda.synchronize()
    for i in range(len(batches)):
        _same_step_outputs(ref[i], staged[i], f"staged {i}")
        _same_step_outputs(ref[i], fetched[i], f"prefetched {i}")
        _same_step_outputs(ref[i], fetched2[i], f"prefetched, no sync {i}")
    # the harness-visible scene entries of a staged batch equal the device-side conversion's
    a, b = wm.pre_processing(batches[0]), wm.pre_processing(_dev_batch(batches[0]))
    for key in b:
        if torch.is_tensor(b[key]) and not key.endswith(("_attr", "_pe")):
            assert a[key].dtype == b[key].dtype and torch.equal(a[key], b[key]), key
    # ---- validation (ground truth travels in the same slab; reactive replay + joint future prediction + metric states)
    wmv = _engine({"time_step_end": 40, "n_joint_future": 2}, sd)
    vb = [synth.make_val_batch(7900 + i, 2, n_agent=20, n_pl=50, n_tl=12, p_future_spawn=0.3, p_future_exit=0.3) for i in range(3)]
    veps = torch.from_numpy(synth.make_latent_noise(6, 2 * 2, 20)).cuda()
    vref = [wmv.validation_step(_dev_batch(b), latent_eps=veps, generator=gen(i)) for i, b in enumerate(vb)]
    vst = [wmv.validation_step(sb, latent_eps=veps, generator=gen(i)) for i, sb in enumerate(wmv.prefetch(vb))]
    torch.cuda.synchronize()
    for x, y in zip(vref, vst):
        for part in ("reactive_replay", "joint_future_pred"):
            assert torch.equal(x[part]["rollout_buffer"].preds, y[part]["rollout_buffer"].preds), part
            assert torch.equal(x[part]["rollout_buffer"].valid, y[part]["rollout_buffer"].valid), part
            assert torch.equal(x[part]["metric_states"], y[part]["metric_states"]), part
            for k_, v in x[part]["pred_dict"].items():
                if torch.is_tensor(v):
                    assert torch.equal(v, y[part]["pred_dict"][k_]), (part, k_)
        assert torch.equal(x["reactive_replay"]["train_states"], y["reactive_replay"]["train_states"])
        assert torch.equal(x["latent_post"].mean, y["latent_post"].mean)


def test_two_lane_pipeline_equals_plain_steps_and_reruns_an_overflowing_batch():
    """`wm.pipeline(loader, lanes=2)`: consecutive batches on two contexts / streams, results in order.  (1) Bit-identical to plain
    `test_step` calls over a stream of distinct batches (K = 2, drawn destinations; the L2 warmers switch themselves off while the
    second context is active -- hints, no numerical effect).  (2) Nothing unchecked leaves the iterator: with weights scaled so that
    an activation leaves the fp16-pair range, every batch comes out re-run on the exact-fp32 kernels (finite, equal to a plain
    checked call) and the pipeline counts the re-runs."""
    import warnings

    from trafficbots_amd import synth

    sd = synth.make_state_dict(5)
    scene = dict(n_agent=20, n_pl=50, n_tl=12, p_late_spawn=0.3, p_invalid_agent=0.2, pos_range=140.0)
    batches = [synth.make_batch(8800 + i, 3, **scene) for i in range(7)]
    k, step_end = 2, 30
    eps = torch.from_numpy(synth.make_latent_noise(5, 3 * k, 20)).cuda()
    wm = _engine({"time_step_end": step_end, "n_joint_future": k}, sd)
    kw = lambda i: dict(latent_eps=eps, generator=torch.Generator(device="cuda").manual_seed(300 + i))  # noqa: E731
    ref = [wm.test_step(b, **kw(i)) for i, b in enumerate(batches)]
    pipe = wm.pipeline(batches, lanes=2, kwargs_fn=kw)
    got = list(pipe)
    torch.cuda.synchronize()
    assert len(got) == len(batches) and pipe.n_reruns == 0
    for i in range(len(batches)):
        _same_step_outputs(ref[i], got[i], f"lane pipeline {i}")
    got3 = list(wm.pipeline(batches, lanes=3, kwargs_fn=kw))
    torch.cuda.synchronize()
    for i in range(len(batches)):
        _same_step_outputs(ref[i], got3[i], f"three lanes {i}")
    # ---- validation_step through two lanes: same per-batch results, and ONE set of accumulated metric states (the lanes' holders merge)
    vb = [synth.make_val_batch(8900 + i, 2, n_agent=20, n_pl=50, n_tl=12, p_future_spawn=0.3, p_future_exit=0.3) for i in range(5)]
    veps = torch.from_numpy(synth.make_latent_noise(6, 2 * 2, 20)).cuda()
    vkw = lambda i: dict(latent_eps=veps, generator=torch.Generator(device="cuda").manual_seed(400 + i))  # noqa: E731
    wa = _engine({"time_step_end": 40, "n_joint_future": 2}, sd)
    vref = [wa.validation_step(b, **vkw(i)) for i, b in enumerate(vb)]
    wb = _engine({"time_step_end": 40, "n_joint_future": 2}, sd)
    vgot = list(wb.pipeline(vb, lanes=2, step="validation_step", kwargs_fn=vkw))
    torch.cuda.synchronize()
    for x, y in zip(vref, vgot):
        for part in ("reactive_replay", "joint_future_pred"):
            assert torch.equal(x[part]["rollout_buffer"].preds, y[part]["rollout_buffer"].preds), part
            assert torch.equal(x[part]["metric_states"], y[part]["metric_states"]), part
    for ha, hb in zip(wa._metric_holders(), wb._metric_holders()):
        assert torch.allclose(ha.states, hb.states, rtol=1e-12, atol=0), type(ha).__name__  # (float64 sums in another order)
    # ---- (2) an overflow inside the pipeline
    big = _scaled(sd, 3e5)  # (the as2pl FFN hidden activations leave the fp16-pair range: test_fp16_pair_operand_range's "flag" case)
    probe = _engine({"time_step_end": 12, "n_joint_future": 1}, big)
    with warnings.catch_warnings(record=True) as wlist:
        warnings.simplefilter("always")
        plain = probe.test_step(batches[0])
    if not any("exact-fp32" in str(w.message) for w in wlist):
        pytest.skip("the scaled weights did not trip the range guard on this build")
    wm2 = _engine({"time_step_end": 12, "n_joint_future": 1}, big)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        pipe2 = wm2.pipeline(batches[:4], lanes=2)
        outs = list(pipe2)
    torch.cuda.synchronize()
    assert pipe2.n_reruns >= 1
    assert all(torch.isfinite(o["rollout_buffer"].preds).all() for o in outs)
    assert torch.equal(outs[0]["rollout_buffer"].preds, plain["rollout_buffer"].preds)


def test_one_outlier_batch_costs_one_rerun_not_the_context():
    """Round 6 ("a checkpoint that trips the range guard halves throughput").  With in-range weights and ONE batch
    that leaves the fp16-pair range -- here a corrupt record: an agent of 3e8 m length, the first encoder layer's output is then a
    GEMM operand of ~1e7 -- the default `fallback_policy` re-runs THAT batch on the exact-fp32 kernels (bit-identical to a context
    created with operand_precision="fp32_exact") and returns to the fp16-pair kernels (`tb_precision_restore`): the batches after it
    are bit-identical to an undisturbed context's and raise nothing.  "sticky" keeps the old behaviour.  Through `wm.pipeline` the
    outlier is re-run too; a neighbour lane's batch in flight at that moment may be re-run conservatively (the flag word belongs to
    the device): its result then comes from the exact kernels -- valid, not bit-identical -- and every lane ends on the fast kernels."""
    import warnings

    from trafficbots_amd import synth

    sd = synth.make_state_dict(5)
    scene = dict(n_agent=20, n_pl=50, n_tl=12, p_late_spawn=0.3, p_invalid_agent=0.2, pos_range=140.0)
    batches = [synth.make_batch(9100 + i, 3, **scene) for i in range(5)]
    bad = dict(batches[1])
    size = np.array(bad["history/agent/size"], copy=True)
    first_valid = int(np.argmax(np.asarray(bad["history/agent/valid"])[0].all(0)))
    size[0, first_valid, 0] = 3e8
    bad["history/agent/size"] = size
    seq = [batches[0], bad, batches[2], batches[3], batches[4]]
    cfg = {"time_step_end": 25, "n_joint_future": 1}
    eps = torch.from_numpy(synth.make_latent_noise(5, 3, 20)).cuda()
    fast = _engine(cfg, sd)
    exact = _engine(dict(cfg, operand_precision="fp32_exact"), sd)
    with warnings.catch_warnings():
        warnings.simplefilter("error", RuntimeWarning)
        ref_fast = {i: fast.test_step(seq[i], latent_eps=eps) for i in (0, 2, 3, 4)}
        ref_exact = {i: exact.test_step(seq[i], latent_eps=eps) for i in range(5)}
    assert torch.isfinite(ref_exact[1]["rollout_buffer"].preds).all()

    wm = _engine(cfg, sd)
    with warnings.catch_warnings():
        warnings.simplefilter("error", RuntimeWarning)
        _same_step_outputs(ref_fast[0], wm.test_step(seq[0], latent_eps=eps), "before the outlier")
    with pytest.warns(RuntimeWarning, match="back on the fp16-pair kernels for the next batch") as rec:
        o1 = wm.test_step(bad, latent_eps=eps)
    # (only the kernel family that overflowed switches: bit-identical to the all-exact context when both did, else as close as the
    # two precisions are to each other)
    both = any("step kernels and the scene encoders" in str(r.message) for r in rec)
    _same_or_close(ref_exact[1], o1, both, "the outlier, re-run")
    st = wm.engine.precision_state()
    assert st["step"] == "fp16_pair" and st["encode"] == "fp16_pair" and not st["activation_overflow"] and wm.n_fallbacks == 1
    with warnings.catch_warnings():
        warnings.simplefilter("error", RuntimeWarning)
        for i in (2, 3, 4):
            _same_step_outputs(ref_fast[i], wm.test_step(seq[i], latent_eps=eps), f"after the outlier {i}")
    assert wm.n_fallbacks == 1

    ws = _engine(cfg, sd)
    ws.fallback_policy = "sticky"
    with pytest.warns(RuntimeWarning, match="this context stays on them"):
        _same_or_close(ref_exact[1], ws.test_step(bad, latent_eps=eps), both, "sticky: the outlier")
    sts = ws.engine.precision_state()
    assert sts["activation_overflow"] and "fp32_exact" in (sts["step"], sts["encode"])
    _same_or_close(ref_exact[2], ws.test_step(seq[2], latent_eps=eps), both, "sticky: after the outlier")

    # the C call by itself: nothing to restore on a context that never fell back / one created on the exact kernels
    assert fast.engine.precision_restore() is False and exact.engine.precision_restore() is False
    assert ws.engine.precision_restore() is True
    sts = ws.engine.precision_state()
    assert sts["step"] == "fp16_pair" and sts["encode"] == "fp16_pair" and not sts["activation_overflow"]

    wp = _engine(cfg, sd)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        pipe = wp.pipeline(seq, lanes=2, kwargs_fn=lambda i: dict(latent_eps=eps))
        outs = list(pipe)
    torch.cuda.synchronize()
    assert 1 <= pipe.n_reruns <= 3 and len(pipe.notes) == pipe.n_reruns and all("back on the fp16-pair" in n for n in pipe.notes)
    _same_or_close(ref_exact[1], outs[1], both, "pipeline: the outlier")
    n_fast = 0
    for i in (0, 2, 3, 4):
        is_fast = torch.equal(outs[i]["rollout_buffer"].preds, ref_fast[i]["rollout_buffer"].preds)
        n_fast += int(is_fast)
        if is_fast:
            _same_step_outputs(ref_fast[i], outs[i], f"pipeline {i}")
        else:
            _same_or_close(ref_exact[i], outs[i], False, f"pipeline {i} (re-run beside the outlier)")
    assert n_fast >= 4 - (pipe.n_reruns - 1) and n_fast >= 2
    assert all(w.engine.precision_state()["step"] == "fp16_pair" and w.engine.precision_state()["encode"] == "fp16_pair" for w in pipe.wms)


def _same_or_close(ref, got, bitwise, what):
    if bitwise:
        return _same_step_outputs(ref, got, what)
    ra, rb = ref["rollout_buffer"], got["rollout_buffer"]
    assert torch.equal(ra.valid, rb.valid), what
    assert torch.isfinite(rb.preds).all(), what
    d = float(((ra.preds - rb.preds).abs() * ra.valid.unsqueeze(-1)).max())
    assert d <= 2e-3, (what, d)  # (25 steps: fp16-pair vs exact fp32 is ~1e-5 there; the bound only says "the same rollout")


def test_steps_and_pipelines_hold_no_device_memory_between_batches():
    """Round 6 (tests/probes/gpu_soak.py, profiles/r06_soak.txt).  (1) With the cyclic collector OFF, repeated `test_step` /
    `validation_step` calls do not grow torch's allocated bytes: a batch's device slab and outputs die with their last reference (the
    scene <-> stand-in cycle kept every slab alive until a collection happened to run).  (2) `wm.pipeline` makes its other lanes'
    contexts once per `wm` and reuses them (a context is a weights arena + workspaces: one per call was a leak of hundreds of
    megabytes and a second of host time each)."""
    import gc

    from trafficbots_amd import synth

    sd = synth.make_state_dict(5)
    scene = dict(n_agent=20, n_pl=50, n_tl=12, p_late_spawn=0.3, p_invalid_agent=0.2, pos_range=140.0)
    batches = [synth.make_batch(9300 + i, 3, **scene) for i in range(3)]
    vb = [synth.make_val_batch(9400 + i, 2, n_agent=20, n_pl=50, n_tl=12) for i in range(2)]
    wm = _engine({"time_step_end": 30, "n_joint_future": 2}, sd)

    def spin(n):
        for i in range(n):
            wm.test_step(batches[i % 3])
            wm.validation_step(vb[i % 2])
        for _ in wm.pipeline([batches[i % 3] for i in range(n)], lanes=2):
            pass
        torch.cuda.synchronize()

    spin(4)
    gc.collect()
    gc.disable()
    try:
        m0 = torch.cuda.memory_allocated()
        clones = [id(c) for c in wm._lane_clones]
        free0 = torch.cuda.mem_get_info()[0]
        spin(12)
        m1 = torch.cuda.memory_allocated()
        free1 = torch.cuda.mem_get_info()[0]
    finally:
        gc.enable()
    assert m1 - m0 <= 64 * 1024, f"torch allocated grew by {(m1 - m0) / 2**20:.2f} MB over 36 steps with the cyclic collector off"
    assert [id(c) for c in wm._lane_clones] == clones and len(clones) == 1
    assert free0 - free1 <= 32 * 2**20, f"device memory in use grew by {(free0 - free1) / 2**20:.1f} MB"


def test_shorter_time_step_current_rolls_out_on_the_whole_history_through_every_feeder():
    """Round 6 (tools/fuzz_oracle_vs_reference.py): with time_step_current = 5 the reference's `test_step` still hands the rollout the
    batch's 11-step history (teacher-forcing mask, state overrides, the validity the kill rule spares: `waymo_motion.py:925-926,
    538-545`) while the encoders see six steps.  Golden `cfg_variant` pins the numbers (tests/test_gpu_parity.py); here: the same
    result through plain calls with a host batch and a device batch, `wm.prefetch` and `wm.pipeline`, and the packed-h5 path refuses
    the configuration instead of rolling out on a truncated history."""
    g, meta = load_golden("cfg_variant")
    cfg, sd, batch, eps = golden_inputs(meta)
    over = {"time_step_end": meta["time_step_end"], "n_joint_future": meta["k"], **meta["overrides"]}
    wm = _engine(over, sd)
    assert wm.n_hist == 6
    e = torch.from_numpy(eps).cuda()
    gs = torch.from_numpy(np.transpose(g["goal_sample"], (0, 2, 1)).copy()).cuda()
    kw = dict(latent_eps=e, goal_sample=gs)
    ref = wm.test_step(batch, **kw)
    # the rule at work: agents valid in the golden beyond step 5 although they sit outside the map (the history still holds them)
    assert torch.equal(ref["rollout_buffer"].valid.cpu(), torch.from_numpy(g["valid"]))
    assert float(g["outside_map"].mean()) > 0.1
    _same_step_outputs(ref, wm.test_step(_dev_batch(batch), **kw), "device batch")
    for i, sb in enumerate(wm.prefetch([batch, batch])):
        assert "hist" in sb
        _same_step_outputs(ref, wm.test_step(sb, **kw), f"prefetch {i}")
    for i, o in enumerate(wm.pipeline([batch, batch, batch], lanes=2, kwargs_fn=lambda i: kw)):
        _same_step_outputs(ref, o, f"pipeline {i}")
    with pytest.raises(NotImplementedError, match="time_step_current = 10"):
        wm.pre_processing({"packed/agent_valid": torch.zeros(1, 6, 2)})
