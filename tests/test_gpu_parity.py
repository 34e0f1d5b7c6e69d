"""GPU parity tests: the HIP path (through the C ABI, via trafficbots_amd.WaymoMotion) against
 (1) the committed goldens of the REFERENCE (tests/golden/*.npz) and (2) the CPU oracle on fresh seeds,
plus size-independent properties at the full BASELINE shape.

Tolerances (fp32, stated per north_star):
  * one-shot products (encoder features, prior mean, destination logits, re-synced one-step policy
    feature): <= 2e-5 abs on O(1) values;
  * open-loop / teacher-forced steps (<= step 10): <= 1e-5 m;
  * closed-loop xy over the 8 s horizon: north_star's target is <= 1e-4 m.  The rollout is chaotic: the REFERENCE's own fp32 run
    differs from its fp64 run by up to 1.3e-4 m at step 90 on the headline shape (fixtures `preds_fp64`; 7.8e-4 m at step 170 of the
    stress shape), so two correct fp32 implementations cannot agree better than that noise.  The noise is MEASURED and the acceptance
    rule is ONE rule (round 4; amended once after its first measurement -- the quantisation term -- and pinned by hash since round 5): tools/ensemble.py::closed_loop_rule / suite_rule
    (the text there is the specification).  In short: every closed-loop golden has a sidecar tests/golden/ensg/<name>.npz with 32
    further fp32 runs of the imported reference that are INDEPENDENT of the base run -- channel-re-labelled weights
    (tools/channel_perm.py: every Linear / LayerNorm / attention product sums in another order) on permuted batches -- and per step
        (a) |hip - reference fp64|(t) <= max(1e-4, PB64(t) + q)     (b) |hip - reference fp32|(t) <= max(1e-4, PB32(t) + q)
    with PB = the one-sided log-normal prediction limit (alpha = 1e-3) of the members' distances to that anchor and q = 2^-16 m, one
    fp32 ulp of a coordinate at the maps' extent (the rule's one change after its freeze, from a leave-one-out test on the reference's
    own runs: profiles/r04_rule_calibration.txt) -- no triangle terms, no multipliers (round 3's rule had both) -- plus
        (c) north_star's flat bound where it is attainable: |hip - reference fp32|(t) <= 1e-4 over the WHOLE horizon on the small
            shapes and for every step t <= FLAT_1E4_UNTIL on the others,
    and, over the whole suite (test_suite_level_closed_loop_parity): the number of cases in which HIP ends farther from the fp64 twin
    than every member <= the 99 % binomial quantile, and the geometric mean of HIP / median member (distance to fp64) <= 1.5, so that a
    2x regression turns the suite red even where each per-case bound still holds.
    Measured values per step are written to gpurun_out/parity_report.json.  Discrete outputs (valid / override / kill /
    destination-reached flags) must be EQUAL.  Oracle-checked cases (no golden) measure the same kind of ensemble with the oracle on the
    spot (`_oracle_ensemble`: permuted batches + Oracle(gemm_order_seed=...), 16 members) and apply rules (a) and (b).
"""
import json
import os

import numpy as np
import pytest
import torch

from conftest import ROOT, golden_inputs, load_golden

pytestmark = pytest.mark.gpu

ONE_SHOT_TOL = 2e-5
REPORT = {}


# simulation step up to which |hip - reference fp32| <= 1e-4 m is asserted flat (None = the whole horizon)
FLAT_1E4_UNTIL = {"c1_plumbing": None, "small_k1": None, "degenerate": None, "edge_scenes": None, "val_edge": None, "edge_scenes2": None, "val_edge2": None, "cfg_variant": None, "val_small": None, "val_alt_losses": None,
                  "stoch_actions": None, "val_irrelevant": None,
                  "masks_k3": 60, "headline_2": 60, "headline_k6": 60, "val_masks": 60,
                  "headline_w_normal": 40, "headline_w_sharp": 40, "headline_w_ln_gamma": 40, "action_override": None,
                  "headline_8": 60, "stress_1": 40, "headline_w_trained": 40, "val_trained": 60, "headline_w_ckpt": 30,
                  # far_scene: positions of ~3 km carry an fp32 ulp of 2.4e-4 m -- one differently rounded state update is already beyond
                  # 1e-4 m; flat over the teacher-forced steps only, the ensemble rule after them
                  "far_scene": 11}
SUITE = {}  # case -> tools/ensemble.py::closed_loop_rule output (reference-made ensembles only), judged by the suite-level test


def _closed_loop_check(name, preds, g, step_axis, rep, step_start=1):
    """Assertions (a), (b), (c) of the module docstring for one golden.  `preds` / g["preds"] have the simulation steps on
    `step_axis` and xy/yaw/spd last; returns nothing, fills `rep`."""
    assert "preds_fp64" in g.files, f"{name}: golden without fp64 twin"
    ax = tuple(i for i in range(preds.ndim) if i != step_axis)
    v32 = g["valid"][..., None]
    both = (g["valid"] & g["valid_fp64"])[..., None]
    d32 = (np.abs(preds - g["preds"]) * v32)[..., :2].max(axis=ax)
    d64 = (np.abs(preds.astype(np.float64) - g["preds_fp64"]) * both)[..., :2].max(axis=ax)
    rep["traj_xy_max"] = float(d32.max())
    rep["traj_xy_vs_fp64_max"] = float(d64.max())
    above = np.nonzero(d32 > 1e-4)[0]
    rep["first_step_above_1e-4_vs_fp32"] = int(above[0]) + step_start if above.size else None
    until = FLAT_1E4_UNTIL[name]
    n_flat = d32.shape[0] if until is None else min(d32.shape[0], until - step_start + 1)
    rep["traj_xy_max_flat_window"] = float(d32[:n_flat].max())
    REPORT[name] = rep
    ensg_path = os.path.join(ROOT, "tests", "golden", "ensg", f"{name}.npz")
    if not os.path.exists(ensg_path):  # a horizon that ends inside the teacher-forced steps (c1_plumbing): nothing chaotic to measure
        # (no ensemble: only cases that meet north_star's flat 1e-4 m over their WHOLE horizon against both reference runs -- c1_plumbing
        # ends inside the teacher-forced steps, edge_scenes / val_edge run 40 / 50 steps)
        assert name in ("c1_plumbing", "edge_scenes", "val_edge", "edge_scenes2", "val_edge2", "cfg_variant") and until is None and d32.max() <= 1e-4 and d64.max() <= 1e-4, \
            f"{name}: {d32.max():.3e} / {d64.max():.3e}"
        return
    from tools import ensemble

    e = np.load(ensg_path)
    assert float(e["equiv_fp64"]) < 1e-9 and int(e["flips"]) == 0, f"{name}: sidecar ensemble is not an ensemble of the same function"
    r = ensemble.closed_loop_rule(d32, d64, e["ensg_d32"], e["ensg_d64"], n_flat)
    rep.update({k: v for k, v in r.items()})
    SUITE[name] = r
    assert r["ok_vs_fp64"], f"{name}: {r['final_vs_fp64']:.3e} from fp64, outside the reference ensemble ({r['bound_vs_fp64']:.3e}; first at step {r['first_step_outside_vs_fp64']})"
    assert r["ok_vs_fp32"], f"{name}: {r['final_vs_fp32']:.3e} from fp32, outside the reference ensemble ({r['bound_vs_fp32']:.3e}; first at step {r['first_step_outside_vs_fp32']})"
    assert r["ok_flat"], f"{name}: {d32[:n_flat].max():.3e} > 1e-4 within the first {n_flat} steps"


N_ORACLE_ENSEMBLE = 16


def _oracle_ensemble(run, batch, k=1, eps=None, dest=None, act=None, n_members=N_ORACLE_ENSEMBLE, seed=0):
    """The measured rounding-noise envelope for a case no golden covers: `run(batch, eps, dest, act, gemm_order_seed)` -> {"preds"
    [N,A,S,4] or [B,A,K,S,4] ..., "valid"} is the fp32 ORACLE; it is re-run on `n_members` permuted batches (tools/ensemble.py), each
    with its own summation order inside the Linears too (Oracle(gemm_order_seed=...): the batch permutation alone only re-orders the
    attention sums), and the outputs un-permuted.  Returns the list of members' (preds, valid) as numpy arrays in the base ordering."""
    from tools import ensemble

    members = []
    for i in range(n_members):
        pb, perm = ensemble.permute_batch({k_: np.asarray(v) for k_, v in batch.items()}, 7919 * (seed + 1) + i)
        r = run(pb, None if eps is None else perm.agents_fwd(eps, k), None if dest is None else perm.dest_fwd(dest, k),
                None if act is None else perm.agents_fwd(act, k), 4001 * (seed + 1) + i)
        p, v = r["preds"].numpy(), r["valid"].numpy()
        kk = k if p.shape[0] == perm.n_scene * k else 1  # [N,A,...] per instance or [B,A,K,...] per scene
        members.append((perm.agents_back(p, kk), perm.agents_back(v, kk)))
    return members


def _assert_closed_loop(preds, r32, r64, what, members=None):
    """|hip - oracle fp64|(t) <= max(1e-4, spread64(t)) and |hip - oracle fp32|(t) <= max(1e-4, spread32(t)): the envelope of the golden
    tests with the ensemble (`members`, from _oracle_ensemble) measured on the spot.  Tensors [..., S, 4] with S second to last."""
    assert members, f"{what}: no ensemble"
    step_axis = preds.ndim - 2
    p32, p64 = r32["preds"].numpy(), r64["preds"].numpy()
    v32, v64 = r32["valid"].numpy(), r64["valid"].numpy()
    ax = tuple(i for i in range(p32.ndim) if i != step_axis)

    def dist(a, b, v):
        return (np.abs(a.astype(np.float64) - b.astype(np.float64)) * v[..., None])[..., :2].max(axis=ax)

    from tools import ensemble

    ens64 = np.stack([dist(p32, p64, v32 & v64)] + [dist(m, p64, mv & v64) for m, mv in members])
    ens32 = np.stack([dist(m, p32, mv & v32) for m, mv in members])
    r = ensemble.closed_loop_rule(dist(preds, p32, v32), dist(preds, p64, v32 & v64), ens32, ens64)
    REPORT[f"oracle_ensemble/{what}"] = r
    assert r["ok_vs_fp64"], f"{what}: {r['final_vs_fp64']:.3e} from fp64 (oracle ensemble {r['bound_vs_fp64']:.3e}; first at step {r['first_step_outside_vs_fp64']})"
    assert r["ok_vs_fp32"], f"{what}: {r['final_vs_fp32']:.3e} from fp32 (oracle ensemble {r['bound_vs_fp32']:.3e}; first at step {r['first_step_outside_vs_fp32']})"
    return r["final_vs_fp32"], r["final_vs_fp64"], r["bound_vs_fp64"]


def _engine(cfg_overrides, sd):
    from trafficbots_amd.waymo_motion import WaymoMotion

    wm = WaymoMotion(**cfg_overrides)
    wm.load_state_dict(sd)
    return wm


def _run(meta, sd, batch, eps, goal_sample=None, tap_step=-1):
    wm = _engine({"time_step_end": meta["time_step_end"], "n_joint_future": meta["k"], **meta.get("overrides", {})}, sd)
    out = wm.test_step(batch, latent_eps=torch.from_numpy(eps).cuda(), goal_sample=goal_sample, tap_step=tap_step)
    torch.cuda.synchronize()
    return wm, out


def _tap_tol(ref, ens=0.0, ens_xy=None):
    """One-shot tolerance on O(1) values, relative to the reference tensor's magnitude where that is larger, or -- for taps inside
    the closed loop -- what the reference ensemble itself spreads by at that tap (`ens`), scaled up to north_star's trajectory
    tolerance where the ensemble's own trajectory spread at that step (`ens_xy`) is below it: features follow the state, so the feature
    distance that goes with an allowed 1e-4 m is ens * 1e-4 / ens_xy (the same floor the trajectory assertions have)."""
    if ens and ens_xy is not None:
        ens = float(ens) * max(1.0, 1e-4 / max(float(ens_xy), 1e-9))
    return max(ONE_SHOT_TOL * max(1.0, float(np.abs(ref).max())), float(ens))


def _check_all_taps(name, g, meta, wm, sd, batch, eps, gs, rep):
    """EVERY tap the golden stores: `tap{s}/policy_feature` and `tap{s}/agent_feature` from one fused rollout per
    tap step, `tap{s}/hidden` / `state_in` / `agent_valid` from a stepwise rollout (tb_rollout_begin / _step / _state)."""
    from trafficbots_amd.runtime import teacher_forcing_mask

    k, a = meta["k"], meta["scene"]["n_agent"]
    n = meta["n_scene"] * k
    ens = lambda key: float(g[key]) if key in g.files else 0.0  # noqa: E731
    from tools import ensemble

    spread32 = np.maximum.accumulate(g["ens_d32"].max(0)) if "ens_d32" in g.files else None
    bound32 = ensemble.prediction_bound(g["ens_d32"]) if "ens_d32" in g.files else None
    ens_xy = lambda s_: None if spread32 is None else float(spread32[min(s_, len(spread32)) - 1])  # noqa: E731
    # (round 6) the tap ensembles stored in the golden are batch-permutation members, which share the base run's GEMM order and
    # under-state what an independent implementation spreads by (round 4); where the golden has the channel-re-labelled sidecar, a
    # closed-loop tap's tolerance is widened by how much farther THOSE members' trajectory prediction limit reaches at that step
    ensg_path = os.path.join(ROOT, "tests", "golden", "ensg", f"{name}.npz")
    pbg = ensemble.prediction_bound(np.load(ensg_path)["ensg_d32"]) if os.path.exists(ensg_path) else None
    widen = lambda s_: 1.0 if pbg is None or spread32 is None else max(1.0, float(pbg[min(s_, len(pbg)) - 1]) / max(ens_xy(s_), 1e-9))  # noqa: E731
    closed = lambda s_: s_ > wm.hparams["time_step_current"] + 1  # noqa: E731  (the state of step s-1 already came from the policy)
    eps_t = torch.from_numpy(eps).cuda()
    for s_ in meta["tap_steps"]:
        out = wm.test_step(batch, latent_eps=eps_t, goal_sample=gs, tap_step=s_)["rollout_buffer"].taps
        for key, got in (("policy_feature", out["tap_policy_feature"]), ("agent_feature", out["tap_agent_feature"])):
            ref = g[f"tap{s_}/{key}"]
            v = g[f"tap{s_}/agent_valid"][..., None]
            err = float((np.abs(got.cpu().numpy() - ref) * v).max())
            rep[f"tap{s_}/{key}"] = err
            tol = _tap_tol(ref * v, ens(f"ens_tap{s_}/{key}") * widen(s_) if closed(s_) else 0.0, ens_xy(s_))
            if key == "policy_feature" and not closed(s_):  # (round 1/2's bound on the re-synced one-step policy feature)
                # (`headline_w_ckpt`: intermediate activations reach ~30 under the checkpoint-like LayerNorm gains; the reference's own
                # fp32 run is 5.7e-6 from its fp64 run on this tap, independent fp32 oracles 3.2e-6 .. 5.7e-6 from the golden -- measured
                # with the oracle, profiles/r06_experiments.txt item 11 -- so 5e-6 is below that case's fp32 noise: 1.5e-5 there)
                tol = (1.5e-5 if name == "headline_w_ckpt" else 5e-6) * max(1.0, float(np.abs(ref * v).max()))
            assert err <= tol, (name, s_, key, err, tol)
    # stepwise: hidden state + simulator state at every tap step
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
    wm.rollout(feats, latent, gs.reshape(n, a).cuda(), gv, teacher_forcing_mask(scene["agent_valid"].bool()), deterministic_latent=det,
               step_end=meta["time_step_end"], k_futures=k, latent_eps=eps_t, stepwise=True)
    prev = wm.engine.rollout_state()
    for t in range(1, max(meta["tap_steps"], default=0) + 1):
        wm.forward()
        st = wm.engine.rollout_state()
        if t in meta["tap_steps"]:
            v = g[f"tap{t}/agent_valid"]
            assert (prev["agent_valid"].bool().cpu().numpy() == v).all(), (name, t, "agent_valid")
            ref = g[f"tap{t}/state_in"]
            err = float((np.abs(prev["agent_state"].cpu().numpy() - ref) * v[..., None])[..., :2].max())
            rep[f"tap{t}/state_in_xy"] = err
            assert err <= max(1e-4, 0.0 if bound32 is None else float(bound32[max(0, min(t - 1, len(bound32)) - 1)])), (name, t, "state_in", err)
            ref = g[f"tap{t}/hidden"].reshape(3, n, a, 128)
            err = float(np.abs(st["hidden"].cpu().numpy() - ref).max())
            rep[f"tap{t}/hidden"] = err
            # (a tap in front of the closed loop: 2e-5 on |h| <= 1 -- or, where the reference's own permutation ensemble already spreads
            # by more than a third of that at this tap, four times its spread: under `headline_w_ckpt`'s sharp attention the
            # reference's fp32 run is 2.1e-4 from its fp64 run on tap1/hidden and independent fp32 oracles 1.1e-4 .. 2.2e-4 from the
            # golden (ensemble spread 7.4e-5; the members share the base run's GEMM order, an independent implementation does not))
            tol_h = _tap_tol(ref, ens(f"ens_tap{t}/hidden") * widen(t) if closed(t) else 0.0, ens_xy(t))
            if not closed(t):
                tol_h = max(tol_h, 4.0 * ens(f"ens_tap{t}/hidden"))  # (independent oracle seeds reach 3.0x the spread: a third of margin)
            assert err <= tol_h, (name, t, "hidden", err, tol_h)
        prev = st
    wm.finish_rollout()


@pytest.mark.parametrize("name", ["c1_plumbing", "small_k1", "masks_k3", "degenerate", "headline_2", "headline_k6",
                                  "headline_w_normal", "headline_w_sharp", "headline_w_ln_gamma", "headline_8", "stress_1",
                                  "headline_w_trained", "headline_w_ckpt", "edge_scenes", "edge_scenes2", "far_scene", "cfg_variant"])
def test_against_reference_golden(name):
    g, meta = load_golden(name)
    cfg, sd, batch, eps = golden_inputs(meta)
    gs = torch.from_numpy(np.transpose(g["goal_sample"], (0, 2, 1)).copy())  # [B,K,A] -> flattened inside
    tap = meta["tap_steps"][0] if meta["tap_steps"] else -1
    wm, out = _run(meta, sd, batch, eps, goal_sample=gs, tap_step=tap)
    buf = out["rollout_buffer"]
    rep = {}
    # ---- encoders
    if "map_feature" in g.files:
        f = out["input_feature_dict"]
        assert (f["map_feature_valid"].cpu().numpy() == g["map_feature_valid"]).all()
        rep["map_feature"] = float(np.abs(f["map_feature"].cpu().numpy() - g["map_feature"]).max())
        rep["agent_feature_cur"] = float(np.abs(f["agent_feature"][:, -1].cpu().numpy() - g["agent_feature_cur"]).max())
        rep["agent_feature_0"] = float(np.abs(f["agent_feature"][:, 0].cpu().numpy() - g["agent_feature_0"]).max())
        rep["tl_feature_cur"] = float(np.abs(f["tl_feature"][:, -1].cpu().numpy() - g["tl_feature_cur"]).max())
        lg = torch.log_softmax(out["dest_logits"], -1).cpu().numpy()
        fin = np.isfinite(g["dest_logits"])
        assert (np.isfinite(lg) == fin).all(), "destination candidate masks differ"
        rep["dest_logits"] = float(np.abs(np.where(fin, lg - np.where(fin, g["dest_logits"], 0), 0)).max())
    rep["latent_mean"] = float(np.abs(out["latent_mean"].cpu().numpy() - g["latent_mean"]).max())
    assert (out["latent_valid"].cpu().numpy() == g["latent_valid"]).all()
    # the personality the rollout prologue drew on the device (MyDist.sample with the golden's eps; K > 1: stochastic instances)
    rep["latent_sample"] = float(np.abs(buf.latent_sample.cpu().numpy() - g["latent_sample"]).max())
    # deterministic destination of instance 0 equals the reference's unless its top-2 logits nearly tie
    own = out["dest_logits"].argmax(-1).cpu().numpy()
    diff = own != g["goal_sample"][:, :, 0]
    if diff.any() and "dest_logits" in g.files:
        top2 = np.sort(np.where(np.isfinite(g["dest_logits"]), g["dest_logits"], -1e30), -1)[..., -2:]
        assert ((top2[..., 1] - top2[..., 0])[diff] < 1e-4).all(), "argmax destination differs beyond a near-tie"
    rep["goal_log_probs"] = float(np.abs(out["goal_log_probs"].cpu().numpy() - g["goal_log_probs"]).max())
    rep["latent_log_probs"] = float(np.abs(buf.latent_log_probs.cpu().numpy() - g["latent_log_probs"]).max())
    rep["action_log_probs"] = float(np.abs(buf.action_log_probs.cpu().numpy() - g["action_log_probs"]).max())
    for k, v in rep.items():
        # the absolute 2e-5 is stated for O(1) values; a feature tensor whose entries reach tens (checkpoint-like LayerNorm gains:
        # `headline_w_ckpt`, map feature up to 27.8) is held to the same RELATIVE accuracy, 4e-6 of its largest entry -- the
        # reference's own fp32 run is 1.05e-4 (3.8e-6 relative) from its fp64 run on that tensor, independent fp32 oracles 5e-5 from it
        scale = float(np.abs(g[k]).max()) if k in ("map_feature", "agent_feature_cur", "agent_feature_0", "tl_feature_cur") else 0.0
        assert v <= max(ONE_SHOT_TOL, 4e-6 * scale), f"{name}: {k} max-abs {v:.3e} (largest entry {scale:.3g})"
    # ---- every stored tap: policy / agent features, hidden and simulator state at the tap steps, final hidden / state
    if meta["tap_steps"]:
        _check_all_taps(name, g, meta, wm, sd, batch, eps, gs, rep)
    fin_v = g["final_valid"]
    assert (buf.final["final_valid"].bool().cpu().numpy() == fin_v).all()
    ens_fs = float(g["ens_final_state"]) if "ens_final_state" in g.files else 0.0
    rep["final_state_xy"] = float((np.abs(buf.final["final_state"].cpu().numpy() - g["final_state"]) * fin_v[..., None])[..., :2].max())
    ensg_path = os.path.join(ROOT, "tests", "golden", "ensg", f"{name}.npz")
    if os.path.exists(ensg_path):  # (the final state is the last step's post-override state: the closed-loop bound of that step)
        from tools import ensemble

        fs_bound = max(1e-4, float(ensemble.prediction_bound(np.load(ensg_path)["ensg_d32"])[-1]))
    else:
        fs_bound = 1e-4
    assert rep["final_state_xy"] <= fs_bound, (name, rep["final_state_xy"], fs_bound)
    if "final_hidden" in g.files:
        fh = g["final_hidden"].reshape(buf.final["final_hidden"].shape)
        rep["final_hidden"] = float(np.abs(buf.final["final_hidden"].cpu().numpy() - fh).max())
        assert rep["final_hidden"] <= _tap_tol(fh, float(g["ens_final_hidden"]) if "ens_final_hidden" in g.files else 0.0, ens_fs), (name, rep["final_hidden"])
    # ---- discrete outputs must be equal
    assert (buf.valid.cpu().numpy() == g["valid"]).all()
    assert (buf.override_masks.cpu().numpy() == g["override_masks"]).all()
    for k in ("outside_map", "outside_map_this_step", "dest_reached", "dest_reached_this_step"):
        assert (buf.violations[k].cpu().numpy() == g[k]).all(), k
    # ---- trajectories
    preds = buf.preds.cpu().numpy()
    d = np.abs(preds - g["preds"]) * g["valid"][..., None]  # [B,A,K,S,4]
    per_step = d[..., :2].max(axis=(0, 1, 2, 4))
    rep["traj_yaw_max"] = float(d[..., 2].max())
    rep["traj_spd_max"] = float(d[..., 3].max())
    n_open = min(10, per_step.shape[0])
    rep["traj_xy_open_loop_max"] = float(per_step[:n_open].max())
    # (teacher-forced steps: one state update from the reference's own states.  1e-5 m for scenes of WOMD's extent; a scene kilometres
    # across -- `far_scene` -- cannot be held below the fp32 spacing of its coordinates: two ulps of the largest one)
    reach = float(np.abs(g["preds"][..., :2]).max())
    open_tol = 1e-5 if reach < 1000.0 else 2.0 * float(np.spacing(np.float32(reach)))
    assert rep["traj_xy_open_loop_max"] <= open_tol, (rep["traj_xy_open_loop_max"], open_tol)
    _closed_loop_check(name, preds, g, 3, rep)


@pytest.mark.parametrize("name", ["small_k1", "masks_k3", "headline_2", "edge_scenes", "edge_scenes2", "cfg_variant"])
def test_exact_fp32_kernels_against_reference_golden(name):
    """`operand_precision="fp32_exact"` -- the fp32 MFMA step kernel and the fp32-MFMA encoder blocks that a context falls back to when a
    tensor or an activation leaves the fp16-pair range (round 4: a product path, not a development switch) -- against the same reference
    goldens and the same closed-loop rule as the default kernels (per case; the suite-level statistics are the default path's)."""
    g, meta = load_golden(name)
    cfg, sd, batch, eps = golden_inputs(meta)
    gs = torch.from_numpy(np.transpose(g["goal_sample"], (0, 2, 1)).copy())
    wm = _engine({"time_step_end": meta["time_step_end"], "n_joint_future": meta["k"], "operand_precision": "fp32_exact", **meta.get("overrides", {})}, sd)
    st = wm.engine.precision_state()
    assert st["step"] == "fp32_exact" and st["encode"] == "fp32_exact" and not st["weight_out_of_range"]
    out = wm.test_step(batch, latent_eps=torch.from_numpy(eps).cuda(), goal_sample=gs)
    torch.cuda.synchronize()
    buf = out["rollout_buffer"]
    rep = {}
    if "map_feature" in g.files:
        f = out["input_feature_dict"]
        rep["map_feature"] = float(np.abs(f["map_feature"].cpu().numpy() - g["map_feature"]).max())
        rep["agent_feature_cur"] = float(np.abs(f["agent_feature"][:, -1].cpu().numpy() - g["agent_feature_cur"]).max())
    rep["latent_mean"] = float(np.abs(out["latent_mean"].cpu().numpy() - g["latent_mean"]).max())
    rep["goal_log_probs"] = float(np.abs(out["goal_log_probs"].cpu().numpy() - g["goal_log_probs"]).max())
    rep["action_log_probs"] = float(np.abs(buf.action_log_probs.cpu().numpy() - g["action_log_probs"]).max())
    for k, v in rep.items():
        assert v <= ONE_SHOT_TOL, f"{name}: {k} max-abs {v:.3e}"
    assert (buf.valid.cpu().numpy() == g["valid"]).all() and (buf.override_masks.cpu().numpy() == g["override_masks"]).all()
    for k in ("outside_map", "dest_reached"):
        assert (buf.violations[k].cpu().numpy() == g[k]).all(), k
    preds = buf.preds.cpu().numpy()
    d = np.abs(preds - g["preds"]) * g["valid"][..., None]
    assert d[..., :2].max(axis=(0, 1, 2, 4))[:10].max() <= 1e-5
    suite_before, report_before = dict(SUITE), REPORT.get(name)
    try:
        _closed_loop_check(name, preds, g, 3, rep)
    finally:  # (the report entry and the suite-level statistics of `name` stay the default kernels')
        REPORT[f"fp32_exact/{name}"] = REPORT.pop(name, rep)
        if report_before is not None:
            REPORT[name] = report_before
        SUITE.clear()
        SUITE.update(suite_before)


@pytest.mark.parametrize("precision", ["fp32", "fp32_exact"])
def test_sampled_actions_against_reference_golden(precision):
    """(`precision="fp32_exact"`, round 5: sampled actions on the exact-fp32 step kernel -- the kernels a range hit falls back to take
    every call the default kernels take.)
    `deterministic_action=False` (`dynamics.py:77`; `training_deterministic_action` in the reference's config): every step samples its
    action as mean + eps * exp(log_std) and scores it with Normal.log_prob.  Golden `stoch_actions` = the reference's
    joint_future_pred with its rollout switched to sampled actions and the rsample draws replaced by synth.make_action_noise."""
    from trafficbots_amd import synth

    name = "stoch_actions"
    g, meta = load_golden(name)
    cfg, sd, batch, eps = golden_inputs(meta)
    n = meta["n_scene"] * meta["k"]
    n_step = meta["time_step_end"] - cfg["time_step_sim_start"] + 1
    act = torch.from_numpy(synth.make_action_noise(meta["base_seed"] + 77, n, meta["scene"]["n_agent"], n_step)).cuda()
    gs = torch.from_numpy(np.transpose(g["goal_sample"], (0, 2, 1)).copy())
    wm = _engine({"time_step_end": meta["time_step_end"], "n_joint_future": meta["k"], "operand_precision": precision}, sd)
    assert wm.engine.precision_state()["step"] == ("fp16_pair" if precision == "fp32" else "fp32_exact")
    out = wm.test_step(batch, latent_eps=torch.from_numpy(eps).cuda(), goal_sample=gs, action_eps=act)
    torch.cuda.synchronize()
    buf = out["rollout_buffer"]
    rep = {"action_log_probs": float(np.abs(buf.action_log_probs.cpu().numpy() - g["action_log_probs"]).max()),
           "latent_log_probs": float(np.abs(buf.latent_log_probs.cpu().numpy() - g["latent_log_probs"]).max())}
    # log-prob of a sample: -(x - mu)^2 / (2 var) with std = e^-2 amplifies the rounding of x - mu by 1 / var ~ 3e3 (values reach ~ -10)
    assert rep["action_log_probs"] <= 2e-4 and rep["latent_log_probs"] <= ONE_SHOT_TOL, rep
    assert (buf.valid.cpu().numpy() == g["valid"]).all() and (buf.override_masks.cpu().numpy() == g["override_masks"]).all()
    for k in ("outside_map", "outside_map_this_step", "dest_reached", "dest_reached_this_step"):
        assert (buf.violations[k].cpu().numpy() == g[k]).all(), k
    preds = buf.preds.cpu().numpy()
    d = np.abs(preds - g["preds"]) * g["valid"][..., None]
    rep["traj_xy_open_loop_max"] = float(d[..., :10, :2].max())
    assert rep["traj_xy_open_loop_max"] <= 1e-5
    suite_before, report_before = dict(SUITE), REPORT.get(name)
    try:
        _closed_loop_check(name, preds, g, 3, rep)
    finally:
        if precision != "fp32":  # (the report entry and the suite-level statistics of `name` stay the default kernels')
            REPORT[f"{precision}/{name}"] = REPORT.pop(name, rep)
            if report_before is not None:
                REPORT[name] = report_before
            SUITE.clear()
            SUITE.update(suite_before)
    # and the deterministic rollout of the same inputs is a different one (the noise is really applied)
    det = wm.test_step(batch, latent_eps=torch.from_numpy(eps).cuda(), goal_sample=gs)["rollout_buffer"].preds
    assert float((det - buf.preds).abs().max()) > 1e-3
    # the same sampled rollout driven step by step through the stateful forward() (round 4): bit-identical to the fused launches
    from trafficbots_amd.runtime import teacher_forcing_mask

    k, a = meta["k"], meta["scene"]["n_agent"]
    scene = wm.pre_processing(batch)
    scene.pop("gt", None)
    f = wm.model.encode_input_features(scene)
    latent = wm.model.latent_encoder()
    wm.model.goal_manager.pred_goal()
    latent.repeat_interleave_(k, 0)
    detl = torch.zeros(n, a, dtype=torch.bool, device="cuda")
    detl[::k] = True
    feats = dict(scene, map_feature=f["map_feature"], map_feature_valid=f["map_feature_valid"].to(torch.uint8), tl_feature=f["tl_feature"])
    gv = scene["agent_valid"].bool().any(1).repeat_interleave(k, 0)
    wm.rollout(feats, latent, gs.reshape(n, a).cuda(), gv, teacher_forcing_mask(scene["agent_valid"].bool()), deterministic_latent=detl,
               deterministic_action=False, action_eps=act, step_end=meta["time_step_end"], k_futures=k, latent_eps=torch.from_numpy(eps).cuda(),
               stepwise=True)
    with pytest.raises(ValueError):
        wm.forward(deterministic_action=True)  # (the mode is bound when the simulator is opened)
    for _ in range(n_step):
        wm.forward(deterministic_action=False)
    sw = wm.finish_rollout()
    sw.flatten_repeat(k)
    assert torch.equal(sw.preds, buf.preds) and torch.equal(sw.action_log_probs, buf.action_log_probs)


@pytest.mark.parametrize("seed,scene", [
    (9100, dict(n_agent=24, n_pl=40, n_tl=40, p_invalid_agent=0.2, p_late_spawn=0.2, p_early_exit=0.2, p_invalid_pl=0.1,
                p_invalid_node=0.3, pos_range=145.0)),
    (9200, dict(n_agent=5, n_pl=17, n_tl=3)),     # ragged: nothing is a multiple of 16
    (9300, dict(n_agent=33, n_pl=100, n_tl=40, p_tl_valid=1.0)),
    (9400, dict(n_agent=128, n_pl=1024, n_tl=40)),  # BASELINE configs[4] shape (stress): 128 agents, 1024 polylines
])
def test_against_oracle_fresh_seeds(seed, scene):
    """HIP path vs the CPU oracle on inputs no golden covers (ragged sizes, other mask mixes), K=2 with
    sampled latents; destinations are taken from the HIP path's own sampler and fed to the oracle."""
    from oracle.trafficbots_oracle import Oracle
    from trafficbots_amd import synth
    from trafficbots_amd.config import load_model_config

    k, step_end, n_scene = 2, 40, 2
    if scene["n_pl"] >= 1024:
        k, step_end, n_scene = 2, 14, 1  # keep the CPU oracle to a few seconds at the stress shape
    meta = dict(time_step_end=step_end, k=k)
    cfg = load_model_config(overrides={"time_step_end": step_end, "n_joint_future": k})
    sd = synth.make_state_dict(seed)
    batch = synth.make_batch(seed, n_scene, **scene)
    eps = synth.make_latent_noise(seed + 1, n_scene * k, scene["n_agent"])
    gen = torch.Generator(device="cuda").manual_seed(seed)
    wm = _engine({"time_step_end": step_end, "n_joint_future": k}, sd)
    out = wm.test_step(batch, latent_eps=torch.from_numpy(eps).cuda(), generator=gen, tap_step=1)
    torch.cuda.synchronize()
    buf = out["rollout_buffer"]
    dest = out["goal_sample"].transpose(1, 2).reshape(n_scene * k, -1).cpu().numpy()
    with torch.no_grad():
        r = Oracle(sd, cfg, torch.float32, hoist=True).joint_future_pred(batch, k, eps, step_end, dest_override=dest, tap_steps=(1,))
    f = out["input_feature_dict"]
    assert (f["map_feature"].cpu() - r["map_feature"]).abs().max() <= ONE_SHOT_TOL
    assert (out["latent_mean"].cpu() - r["latent_mean"]).abs().max() <= ONE_SHOT_TOL
    lg_m, lg_r = out["dest_logits"].cpu(), r["dest_logits_raw"]
    fin = torch.isfinite(lg_r)
    assert (torch.isfinite(lg_m) == fin).all()
    assert torch.where(fin, lg_m - lg_r, torch.zeros_like(lg_r)).abs().max() <= ONE_SHOT_TOL
    assert (buf.taps["tap_policy_feature"].cpu() - r["tap1/policy_feature"]).abs().max() <= 5e-6
    assert (buf.valid.cpu() == r["valid"]).all()
    assert (buf.override_masks.cpu() == r["override_masks"]).all()
    for key in ("outside_map", "dest_reached"):
        assert (buf.violations[key].cpu() == r[key]).all()
    d = (buf.preds.cpu() - r["preds"]).abs() * r["valid"].unsqueeze(-1)
    assert d[..., :2].max() <= 1e-4, f"xy err {d[..., :2].max():.3e}"


def _headline_batch(n_scene, seed=7000):
    from trafficbots_amd import synth

    return synth.make_batch(seed, n_scene, n_agent=64, n_pl=256, n_tl=40)


def test_full_size_properties():
    """Properties at the BASELINE configs[1] shape (B=32, A=64, P=256, 90 steps), independent of any oracle:
    bitwise run-to-run determinism, batch-composition independence (a scene's result does not depend on its
    neighbours or on K), deterministic sample k=0 of a K=6 run equals the K=1 run, finite outputs,
    teacher-forced steps reproduce the history."""
    from trafficbots_amd import synth

    sd = synth.make_state_dict(7)
    batch = _headline_batch(32)
    eps1 = synth.make_latent_noise(1, 32, 64)
    wm = _engine({"time_step_end": 90, "n_joint_future": 1}, sd)
    a = wm.test_step(batch, latent_eps=torch.from_numpy(eps1).cuda())
    b = wm.test_step(batch, latent_eps=torch.from_numpy(eps1).cuda())
    torch.cuda.synchronize()
    pa, pb = a["rollout_buffer"].preds, b["rollout_buffer"].preds
    assert torch.isfinite(pa).all()
    assert torch.equal(pa, pb), "run-to-run results are not bitwise identical"
    assert torch.equal(a["rollout_buffer"].valid, b["rollout_buffer"].valid)
    # sub-batch (scenes 8..11) alone gives bitwise the same trajectories
    sub = {k_: v[8:12] for k_, v in batch.items()}
    c = wm.test_step(sub, latent_eps=torch.from_numpy(eps1[8:12]).cuda())
    assert torch.equal(c["rollout_buffer"].preds, pa[8:12])
    # teacher-forced warm-up: the state fed back at steps <= 10 is the history, so pred at step t (made from
    # history t-1) stays within a physically plausible distance of history t
    hist = torch.from_numpy(batch["history/agent/pos"]).cuda()  # [B,11,A,2]
    pred_xy = pa[:, :, 0, :10, :2]                                # steps 1..10
    gap = (pred_xy - hist[:, 1:11].transpose(1, 2)).norm(dim=-1)
    assert gap.max() < 3.0
    # K = 6: deterministic sample equals the K = 1 run
    wm6 = _engine({"time_step_end": 90, "n_joint_future": 6}, sd)
    sub2 = {k_: v[:2] for k_, v in batch.items()}
    eps6 = synth.make_latent_noise(2, 12, 64)
    d = wm6.test_step(sub2, latent_eps=torch.from_numpy(eps6).cuda(), generator=torch.Generator(device="cuda").manual_seed(3))
    assert torch.equal(d["rollout_buffer"].preds[:, :, 0], pa[:2, :, 0])
    # the sampled futures do differ
    assert (d["rollout_buffer"].preds[:, :, 1] - d["rollout_buffer"].preds[:, :, 0]).abs().max() > 1e-3


def test_stepwise_forward_equals_fused_rollout():
    """`rollout(stepwise=True)` + repeated `forward()` (tb_rollout_begin / _step / _state) is bitwise the fused
    `tb_rollout`, and the per-step return values are the buffer slices of that step."""
    from trafficbots_amd import synth
    from trafficbots_amd.runtime import teacher_forcing_mask

    step_end, k = 25, 2
    sd = synth.make_state_dict(5)
    batch = synth.make_batch(7700, 3, n_agent=20, n_pl=50, n_tl=12, p_late_spawn=0.3, p_invalid_agent=0.2, pos_range=140.0)
    wm = _engine({"time_step_end": step_end, "n_joint_future": k}, sd)
    eps = torch.from_numpy(synth.make_latent_noise(5, 3 * k, 20)).cuda()
    full = wm.test_step(batch, latent_eps=eps, generator=torch.Generator(device="cuda").manual_seed(9))
    ref_buf = full["rollout_buffer"]
    scene = wm.pre_processing(batch)
    f = wm.model.encode_input_features(scene)
    latent, goal = wm.model.latent_encoder(), wm.model.goal_manager.pred_goal()
    latent.repeat_interleave_(k, 0)
    det = torch.zeros(3 * k, 20, dtype=torch.bool, device="cuda")
    det[::k] = True
    feats = dict(scene, map_feature=f["map_feature"], map_feature_valid=f["map_feature_valid"].to(torch.uint8), tl_feature=f["tl_feature"])
    gs = full["goal_sample"].transpose(1, 2).reshape(3 * k, 20)
    gv = scene["agent_valid"].bool().any(1).repeat_interleave(k, 0)
    buf = wm.rollout(feats, latent, gs, gv, teacher_forcing_mask(scene["agent_valid"].bool()), deterministic_latent=det,
                     step_end=step_end, k_futures=k, latent_eps=eps, stepwise=True)
    for t in range(1, step_end + 1):
        state, valid, train, _ = wm.forward()
        s = t - 1
        assert torch.equal(train["pred_state"], buf.preds[:, :, s])
        if t == 7:
            assert valid.dtype == torch.bool and state.shape == (3 * k, 20, 4)
    torch.cuda.synchronize()
    buf = wm.finish_rollout()
    buf.flatten_repeat(k)
    assert torch.equal(buf.preds, ref_buf.preds)
    assert torch.equal(buf.valid, ref_buf.valid)
    assert torch.equal(buf.violations["dest_reached"], ref_buf.violations["dest_reached"])
    with pytest.raises(RuntimeError):
        wm.forward()  # past step_end


def test_fp32_mfma_step_kernels_match_default(monkeypatch):
    """The default step kernel (k_step_x: fp16-pair operands on the XDL pipe, fp32 accumulate) against the fp32-MFMA
    twin kept selectable with TB_STEP_KERNEL=fp32 (k_step): same arithmetic per
    agent up to rounding order -> equal flags, trajectories within the closed-loop tolerance."""
    from trafficbots_amd import synth

    step_end, k = 50, 2
    sd = synth.make_state_dict(11)
    batch = synth.make_batch(9100, 3, n_agent=40, n_pl=96, n_tl=20, p_late_spawn=0.2, p_invalid_agent=0.2, pos_range=140.0)
    eps = torch.from_numpy(synth.make_latent_noise(11, 3 * k, 40)).cuda()
    outs = {}
    for kern in ("xdl", "fp32"):
        monkeypatch.setenv("TB_STEP_KERNEL", kern)
        wm = _engine({"time_step_end": step_end, "n_joint_future": k}, sd)
        outs[kern] = wm.test_step(batch, latent_eps=eps, generator=torch.Generator(device="cuda").manual_seed(4))["rollout_buffer"]
    a = outs["xdl"]
    for kern in ("fp32",):
        b = outs[kern]
        assert torch.equal(a.valid, b.valid)
        assert torch.equal(a.violations["dest_reached"], b.violations["dest_reached"])
        assert torch.equal(a.violations["outside_map"], b.violations["outside_map"])
        err = (a.preds - b.preds).abs().max().item()
        REPORT[f"k_step_x_vs_{kern}_max_abs"] = err
        assert err <= 1e-4, (kern, err)


@pytest.mark.parametrize("name", ["rules_k2", "rules_passive", "rules_edge"])
def test_rule_checks_kernel(name):
    """SURVEY 8(f)-1, `tb_rule_checks`: (1) on the (valid, state) pairs the REFERENCE handed to TrafficRuleChecker.check the
    HIP kernels must return the reference's flags exactly (compare work only, same operand order); (2) end to end -- flags
    enabled in the config, checks evaluated on the HIP rollout's own states -- the flags may differ from the golden only where
    a trajectory difference of ~1e-5 m moves a box across a threshold: at most 0.5 % of the agent-steps."""
    from trafficbots_amd.runtime import RULE_KEYS, scene_from_batch

    g, meta = load_golden(name)
    cfg, sd, batch, eps = golden_inputs(meta)
    k = meta["k"]
    flags = {f"enable_check_{c}": True for c in ("collided", "run_road_edge", "run_red_light", "passive")}
    wm = _engine({"time_step_end": meta["time_step_end"], "n_joint_future": k, "traffic_rule_checker": flags}, sd)
    scene = scene_from_batch(batch, wm.device)
    res = wm.engine.rule_checks(scene, torch.from_numpy(g["check_state"]).cuda(), torch.from_numpy(g["check_valid"]).cuda(), k, flags)
    torch.cuda.synchronize()
    n_true = 0
    for key in RULE_KEYS:
        ref = np.transpose(g[key], (0, 2, 1, 3)).reshape(res[key].shape)
        got = res[key].cpu().numpy().astype(bool)
        assert np.array_equal(got, ref), (key, int((got != ref).sum()), int(ref.sum()))
        n_true += int(ref.sum())
    assert n_true > 0
    # disabled checks return zeros (traffic_rule_checker.py:430, 438, 456, 472)
    off = wm.engine.rule_checks(scene, torch.from_numpy(g["check_state"]).cuda(), torch.from_numpy(g["check_valid"]).cuda(), k,
                                {"enable_check_collided": True})
    assert int(off["run_road_edge"].sum()) == 0 and int(off["passive"].sum()) == 0 and int(off["run_red_light_this_step"].sum()) == 0
    assert np.array_equal(off["collided"].cpu().numpy(), res["collided"].cpu().numpy())
    # end to end
    gs = torch.from_numpy(np.transpose(g["goal_sample"], (0, 2, 1)).copy())
    out = wm.test_step(batch, latent_eps=torch.from_numpy(eps).cuda(), goal_sample=gs)
    buf = out["rollout_buffer"]
    torch.cuda.synchronize()
    assert torch.equal(buf.valid.cpu(), torch.from_numpy(g["valid"]))
    rep = {}
    for key in RULE_KEYS:
        got = buf.violations[key].cpu().numpy()
        diff = int((got != g[key]).sum())
        rep[key] = {"true_ref": int(g[key].sum()), "mismatch": diff}
        assert diff <= 0.005 * got.size, (key, diff)
    REPORT[name] = rep


def test_post_processing_kernel():
    """SURVEY 8(f)-2, `tb_post_process` against the goldens of the reference's WaymoPostProcessing: the same modes selected
    (top-k / MTR NMS), scores within 1e-6 (MPA NMS, temperature softmax), trajectories exact copies, [B,S,A,K] layout."""
    import json

    from conftest import GOLDEN_DIR
    from trafficbots_amd import synth
    from trafficbots_amd.post_processing import WaymoPostProcessing
    from trafficbots_amd.runtime import HipEngine
    from trafficbots_amd.config import load_model_config

    g = np.load(os.path.join(GOLDEN_DIR, "post_processing.npz"))
    meta = json.loads(bytes(g["meta_json"]).decode())
    eng = HipEngine(load_model_config())
    for i, (name, c) in enumerate(meta["cases"].items()):
        valid, scores, trajs, agent_type = synth.make_post_inputs(meta["seed0"] + i, meta["n_scene"], meta["n_agent"], c["n_pred"], meta["n_step"])
        pp = WaymoPostProcessing(eng, c["k_pred"], c["score_temperature"], c["mpa"], c["mtr"], c["aggr"], c["n_iter_em"], c["use_ade"])
        out = pp(torch.from_numpy(valid), torch.from_numpy(scores), torch.from_numpy(trajs), torch.from_numpy(agent_type))
        torch.cuda.synchronize()
        idx, sc = out["mode_idx"].cpu().numpy().astype(np.int64), out["waymo_scores"].cpu().numpy()
        ref_idx, ref_s = g[f"{name}/mode_idx"].astype(np.int64), g[f"{name}/waymo_scores"]
        o_ref, o_got = np.argsort(ref_idx, -1), np.argsort(idx, -1)
        assert np.array_equal(np.take_along_axis(ref_idx, o_ref, -1), np.take_along_axis(idx, o_got, -1)), name
        err = np.abs(np.take_along_axis(ref_s, o_ref, -1) - np.take_along_axis(sc, o_got, -1)).max()
        REPORT[f"post_processing/{name}/score_max_abs"] = float(err)
        assert err < 1e-6, (name, err)
        assert np.array_equal(out["waymo_valid"].cpu().numpy(), g[f"{name}/waymo_valid"])
        sel = np.moveaxis(np.take_along_axis(trajs, idx[..., None, None], 2), 3, 1)  # [B,S,A,K,4]
        assert np.array_equal(out["waymo_trajs"].cpu().numpy(), sel[..., :2])
        assert np.array_equal(out["waymo_yaw_bbox"].cpu().numpy(), sel[..., 2:3])
        assert np.array_equal(out["waymo_spd"].cpu().numpy(), sel[..., 3:4])
    with pytest.raises(NotImplementedError):
        WaymoPostProcessing(eng, aggr_thresh=[2.0])


def test_bf16_operand_mode():
    """`operand_precision="bf16"` (tb_config.operand_precision = 1, BASELINE.json configs 4/5): the same kernels with one bf16
    plane per MFMA operand and fp32 accumulation.  No fp32-parity claim: bf16 rounds every GEMM / attention operand to 8 bits, so
    only short-horizon quantities are compared, with the tolerance bf16 allows -- the re-synced one-step policy feature within 5e-2
    of the reference (values are O(1)), the first free-running step within 2 cm, everything finite, teacher-forced bookkeeping equal."""
    g, meta = load_golden("small_k1")
    cfg, sd, batch, eps = golden_inputs(meta)
    from trafficbots_amd.waymo_motion import WaymoMotion

    wm = WaymoMotion(time_step_end=meta["time_step_end"], n_joint_future=meta["k"], operand_precision="bf16")
    wm.load_state_dict(sd)
    gs = torch.from_numpy(np.transpose(g["goal_sample"], (0, 2, 1)).copy())
    out = wm.test_step(batch, latent_eps=torch.from_numpy(eps).cuda(), goal_sample=gs, tap_step=1)
    torch.cuda.synchronize()
    buf = out["rollout_buffer"]
    preds = buf.preds.cpu().numpy()
    assert np.isfinite(preds).all()
    s_free = meta_step = 11 - 1  # buffer slot of step 11, the first free-running step
    assert np.array_equal(buf.valid.cpu().numpy()[..., : s_free + 1], g["valid"][..., : s_free + 1])
    assert np.array_equal(buf.override_masks.cpu().numpy(), g["override_masks"])
    v = g["valid"][..., s_free]
    err_xy = np.abs(preds[..., s_free, :2] - g["preds"][..., s_free, :2])[v].max()
    tap = buf.taps["tap_policy_feature"].cpu().numpy()
    ref = g["tap1/policy_feature"]
    err_f = np.abs(tap - ref).max()
    REPORT["bf16_mode"] = {"policy_feature_step1_max_abs": float(err_f), "xy_first_free_step_max_abs": float(err_xy),
                           "xy_final_step_max_abs": float(np.abs(preds[..., -1, :2] - g["preds"][..., -1, :2])[g["valid"][..., -1]].max())}
    assert err_f <= 5e-2, err_f
    assert err_xy <= 2e-2, err_xy
    with pytest.raises(ValueError):
        WaymoMotion(operand_precision="fp8")


def test_metric_partials_kernel():
    """`tb_metric_partials` against the states the reference's ErrorMetrics / TrafficRuleMetrics accumulated (metrics.npz): the nine
    counters exactly, the three error sums to 1e-5 relative (fp32 summation order)."""
    import json

    from conftest import GOLDEN_DIR
    from trafficbots_amd import synth
    from trafficbots_amd.config import load_model_config
    from trafficbots_amd.runtime import METRIC_FIELDS, HipEngine

    g = np.load(os.path.join(GOLDEN_DIR, "metrics.npz"))
    meta = json.loads(bytes(g["meta_json"]).decode())
    assert tuple(meta["fields"]) == METRIC_FIELDS
    eng = HipEngine(load_model_config())
    for name, c in meta["cases"].items():
        d = {k: torch.from_numpy(v) for k, v in synth.make_metric_inputs(c["seed"], c["n_scene"], c["n_agent"], c["k"], c["n_step"]).items()}
        vio = {k: d[k] for k in ("outside_map", "collided", "run_road_edge", "run_red_light", "passive", "goal_reached", "dest_reached")}
        out = eng.metric_partials(d["pred_valid"], d["pred_states"], d["override_masks"], vio, d["agent_type"], d["agent_role"],
                                  d["gt_valid"], d["gt_states"], c["tf"]).cpu().numpy()
        ref = g[name]
        counters = [0, 4, 5, 6, 7, 8, 9, 10, 11, 12]
        assert np.array_equal(out[counters], ref[counters]), (name, out, ref)
        assert np.allclose(out[1:4], ref[1:4], rtol=1e-5, atol=0), (name, out[1:4], ref[1:4])
        # without ground truth the error sums stay zero and the rule sums do not change
        out2 = eng.metric_partials(d["pred_valid"], d["pred_states"], d["override_masks"], vio, d["agent_type"], d["agent_role"],
                                   None, None, c["tf"]).cpu().numpy()
        assert np.array_equal(out2[:4], np.zeros(4)) and np.array_equal(out2[4:], out[4:])


TRAIN_FIELDS = ("vae_kl_counter", "vae_kl", "diffbar_reward_counter", "diffbar_reward", "goal_loss", "goal_counter")


@pytest.mark.parametrize("name", ["val_small", "val_masks", "val_alt_losses", "val_irrelevant", "val_trained", "val_edge", "val_edge2"])
def test_validation_step_against_reference_golden(name):
    """SURVEY 8(f)-3 through the C ABI (tb_encode_posterior, tb_rollout driven by the 91-step ground truth, tb_train_partials,
    tb_rule_checks' goal_reached, tb_metric_partials) against what the imported reference's validation_step produced."""
    from trafficbots_amd import synth

    g, meta = load_golden(name)
    over = {"time_step_end": meta["time_step_end"], "n_joint_future": 2}
    over.update(meta["overrides"])
    wm = _engine(over, synth.case_state_dict(meta))
    batch = synth.make_val_batch(meta["base_seed"], meta["n_scene"], **meta["scene"])
    irr = torch.from_numpy(g["irrelevant_draw"]).cuda() if "irrelevant_draw" in g.files else None  # (p_loss_for_irrelevant > 0)
    out = wm.validation_step(batch, irrelevant_draw=irr)
    torch.cuda.synchronize()
    rr = out["reactive_replay"]
    buf = rr["rollout_buffer"]  # flattened [B,A,1,S,...]
    rep = {}
    rep["post_mean"] = float(np.abs(out["latent_post"].mean.cpu().numpy() - g["post_mean"]).max())
    rep["prior_mean"] = float(np.abs(out["latent_prior_mean"].cpu().numpy() - g["prior_mean"]).max())
    assert (out["latent_post"].valid.cpu().numpy() == g["post_valid"]).all()
    assert (out["latent_prior_valid"].cpu().numpy() == g["prior_valid"]).all()
    rep["latent_log_probs"] = float(np.abs(buf.latent_log_probs[:, :, 0].cpu().numpy() - g["latent_log_probs"]).max())
    for k, v in rep.items():
        assert v <= ONE_SHOT_TOL, f"{name}: {k} max-abs {v:.3e}"
    # discrete outputs of the replay must be equal: spawns from ground truth up to step 90, kills, reached flags
    assert (buf.valid[:, :, 0].cpu().numpy() == g["valid"]).all()
    assert (buf.override_masks[:, :, 0].cpu().numpy() == g["override_masks"]).all()
    for k in ("outside_map", "dest_reached", "goal_reached"):
        assert (buf.violations[k][:, :, 0].cpu().numpy() == g[k]).all(), k
    assert (buf.diffbar_rewards_valid[:, :, 0].cpu().numpy() == g["diffbar_rewards_valid"]).all()
    preds = buf.preds[:, :, 0].cpu().numpy()
    d = np.abs(preds - g["preds"]) * g["valid"][..., None]
    _closed_loop_check(name, preds, g, 2, rep)
    # rewards follow the trajectories: the criteria are 1-Lipschitz (SmoothL1 / L1) or quadratic (MSE) in errors of <= 2.5e-4
    rw = buf.diffbar_rewards[:, :, 0].cpu().numpy()
    rep["reward_max_abs"] = float(np.abs(rw - g["diffbar_rewards"]).max())
    assert np.allclose(rw, g["diffbar_rewards"], rtol=5e-5, atol=5e-4)
    ts = rr["train_states"].cpu().numpy()
    for i in (0, 2, 5):
        assert ts[i] == g["train_states"][i], TRAIN_FIELDS[i]
    assert np.allclose(ts, g["train_states"], rtol=1e-4), (ts, g["train_states"])
    comp = wm.train_metrics_reactive_replay.compute()
    ref = json.loads(bytes(g["train_compute_json"]).decode())
    assert set(comp) == set(ref) and all(abs(comp[k] - ref[k]) <= 1e-4 * max(1.0, abs(ref[k])) for k in ref), (comp, ref)
    ms = rr["metric_states"].cpu().numpy()
    want = np.concatenate([g["err_states"], g["rule_states"]])
    assert (ms[[0, 4, 5, 6, 7, 8, 9, 10, 11, 12]] == want[[0, 4, 5, 6, 7, 8, 9, 10, 11, 12]]).all(), (ms, want)
    assert np.allclose(ms, want, rtol=1e-4)
    rep["loss"] = comp["reactive_replay/loss"]
    # the loss kernels on the REFERENCE's buffer: their own arithmetic, no trajectory noise
    gt = wm.pre_processing(batch)["gt"]
    gv, gs = wm._gt_slices(gt, wm.hparams["time_step_sim_start"], meta["time_step_end"])
    raw = {k: torch.from_numpy(g[k]).cuda() for k in ("valid", "preds", "override_masks")}
    rw2, rv2, st2 = wm.engine.train_partials(
        raw, gv, gs, gt["agent_size"], dest_logits=out["dest_logits"], goal_valid=wm.pre_processing(batch)["agent_valid"].bool().any(1),
        gt_dest=gt["gt_dest"],
        post={"latent_mean": torch.from_numpy(g["post_mean"]).cuda(), "latent_valid": torch.from_numpy(g["post_valid"]).cuda()},
        prior={"latent_mean": torch.from_numpy(g["prior_mean"]).cuda(), "latent_valid": torch.from_numpy(g["prior_valid"]).cuda()},
        agent_role=gt["agent_role"], irrelevant_draw=irr)
    torch.cuda.synchronize()
    assert (rv2.bool().cpu().numpy() == g["diffbar_rewards_valid"]).all()
    assert np.allclose(rw2.cpu().numpy(), g["diffbar_rewards"], rtol=3e-6, atol=3e-6)
    assert np.allclose(st2.cpu().numpy(), g["train_states"], rtol=2e-5)
    # second half of validation_step ran too (prior samples, predicted destinations, kill rule against the ground truth)
    bj = out["joint_future_pred"]["rollout_buffer"]
    assert bj.valid.shape[2] == 2 and out["joint_future_pred"]["pred_dict"]["waymo_trajs"].shape[3] == 2
    REPORT[name] = rep


def test_validation_step_against_oracle_fresh_seed():
    """validation_step at a shape no golden covers (40 agents, 100 polylines, ragged), collision reward on, all four rule checks on
    (so the replay's checker reads the 91-step traffic lights), against the CPU oracles."""
    from oracle import training_oracle as TO
    from oracle.rule_checks_oracle import rule_checks
    from oracle.trafficbots_oracle import Oracle
    from trafficbots_amd import synth
    from trafficbots_amd.config import load_model_config

    over = {"time_step_end": 90, "n_joint_future": 1, "differentiable_reward.w_collision": 0.3,
            "traffic_rule_checker": {"enable_check_collided": True, "enable_check_run_road_edge": True,
                                     "enable_check_run_red_light": True, "enable_check_passive": True}}
    scene = dict(n_agent=40, n_pl=100, n_tl=40, p_invalid_agent=0.2, p_late_spawn=0.2, p_future_spawn=0.5, p_future_exit=0.3,
                 pos_range=35.0, p_tl_valid=0.6)
    sd = synth.make_state_dict(9700)
    batch = synth.make_val_batch(9701, 2, **scene)
    wm = _engine(over, sd)
    out = wm.validation_step(batch)
    torch.cuda.synchronize()
    cfg = load_model_config(overrides=over)
    r = Oracle(sd, cfg, dtype=torch.float32).reactive_replay(batch, 90)
    buf = out["reactive_replay"]["rollout_buffer"]
    assert np.abs(out["latent_post"].mean.cpu().numpy() - r["post_mean"].numpy()).max() <= ONE_SHOT_TOL
    assert (out["latent_post"].valid.cpu().numpy() == r["post_valid"].numpy()).all()
    for k in ("valid", "override_masks"):
        assert (getattr(buf, k)[:, :, 0].cpu().numpy() == r[k].numpy()).all(), k
    for k in ("outside_map", "dest_reached", "goal_reached"):
        assert (buf.violations[k][:, :, 0].cpu().numpy() == r[k].numpy()).all(), k
    r64 = Oracle(sd, cfg, dtype=torch.float64).reactive_replay(batch, 90)
    mem = _oracle_ensemble(lambda pb, e_, d_, a_, g_: Oracle(sd, cfg, dtype=torch.float32, gemm_order_seed=g_).reactive_replay(pb, 90), batch, seed=1)
    _assert_closed_loop(buf.preds[:, :, 0].cpu().numpy(), r, r64, "reactive replay", mem)
    gv = r["gt_valid"][:, 1:91].transpose(1, 2)
    gs = r["gt_state"][:, 1:91].transpose(1, 2)
    # losses: oracle arithmetic on the HIP buffer (isolates the loss kernels from trajectory noise)
    pv, ps = buf.valid[:, :, 0].cpu(), buf.preds[:, :, 0].cpu()
    rew, rv = TO.differentiable_reward(pv, ps, gv, gs, r["agent_size"], cfg["differentiable_reward"])
    assert (buf.diffbar_rewards_valid[:, :, 0].cpu() == rv).all()
    assert np.allclose(buf.diffbar_rewards[:, :, 0].cpu().numpy(), rew.numpy(), rtol=5e-6, atol=5e-6)
    st = TO.training_metric_states(pv, rv, rew, buf.override_masks[:, :, 0].cpu(), r["agent_role"], out["dest_logits"].cpu(), r["goal_valid"],
                                   r["gt_dest"], out["latent_post"].mean.cpu(), r["post_log_std"], out["latent_post"].valid.cpu(),
                                   out["latent_prior_mean"].cpu(), r["prior_log_std"], out["latent_prior_valid"].cpu(),
                                   cfg["training_metrics"])
    ts = out["reactive_replay"]["train_states"].cpu().numpy()
    assert np.allclose(ts, np.array([st[k] for k in TRAIN_FIELDS]), rtol=2e-5), (ts, st)
    # flag-gated checks of the replay: oracle on the recorded per-step states with the GROUND-TRUTH traffic lights
    inp = r["_inp"]
    b64 = {k: torch.from_numpy(np.asarray(batch[k])) for k in ("tl_stop/valid", "tl_stop/pos", "tl_stop/state")}
    ref = rule_checks(r["check_state"], r["check_valid"], 1, 1, inp["agent_type"], inp["agent_size"], inp["map_valid"], inp["map_type"],
                      inp["map_pos"], inp["map_dir"], b64["tl_stop/valid"], b64["tl_stop/pos"], b64["tl_stop/state"])
    for k, want in ref.items():
        got = buf.violations[k][:, :, 0].cpu().numpy()
        assert (got == want.numpy()).mean() >= 0.999, k  # (states differ by <= 2.5e-4 m: a threshold can flip on a rare tie)
    assert ref["collided"].any() and ref["run_road_edge"].any()
    # second half of validation_step: joint_future_pred against the ground truth (kill rule, goal_reached), K = 1 deterministic
    bj = out["joint_future_pred"]["rollout_buffer"]
    dest = out["joint_future_pred"]["goal_sample"].transpose(1, 2).reshape(2, -1).cpu().numpy()
    rj = Oracle(sd, cfg, dtype=torch.float32).joint_future_pred(batch, 1, None, 90, dest_override=dest, use_gt=True)
    for k in ("valid", "override_masks"):
        assert (getattr(bj, k).cpu().numpy() == rj[k].numpy()).all(), k
    for k in ("outside_map", "dest_reached", "goal_reached"):
        assert (bj.violations[k].cpu().numpy() == rj[k].numpy()).all(), k
    rj64 = Oracle(sd, cfg, dtype=torch.float64).joint_future_pred(batch, 1, None, 90, dest_override=dest, use_gt=True)
    mem = _oracle_ensemble(lambda pb, e_, d_, a_, g_: Oracle(sd, cfg, dtype=torch.float32, gemm_order_seed=g_).joint_future_pred(pb, 1, None, 90, dest_override=d_, use_gt=True),
                           batch, dest=dest, seed=2)
    _assert_closed_loop(bj.preds.cpu().numpy(), rj, rj64, "validation joint_future_pred", mem)


@pytest.mark.parametrize("rollout_prior,sampled_actions", [(False, False), (True, False), (False, True)])
def test_training_step_forward_against_oracle(rollout_prior, sampled_actions):
    """Forward value of training_step (teacher_forcing_training, a SAMPLE of the posterior / prior personality, ground-truth
    destination, TrainingMetrics.compute) against the oracles; `sampled_actions` = the reference's
    training_deterministic_action: false (`waymo_motion.py:398`) with explicit draws."""
    from oracle import training_oracle as TO
    from oracle.trafficbots_oracle import Oracle
    from trafficbots_amd import synth
    from trafficbots_amd.config import load_model_config

    over = {"time_step_end": 60, "differentiable_reward.w_collision": 0.2, "training_deterministic_action": not sampled_actions}
    scene = dict(n_agent=20, n_pl=48, n_tl=12, p_invalid_agent=0.2, p_late_spawn=0.3, p_future_spawn=0.5, p_future_exit=0.3, pos_range=60.0)
    sd = synth.make_state_dict(9800)
    batch = synth.make_val_batch(9801, 3, **scene)
    eps = synth.make_latent_noise(9802, 3, 20)
    act = synth.make_action_noise(9803, 3, 20, 60) if sampled_actions else None
    wm = _engine(over, sd)
    out = wm.training_step(batch, latent_eps=torch.from_numpy(eps).cuda(), rollout_prior=rollout_prior,
                           action_eps=None if act is None else torch.from_numpy(act).cuda())
    torch.cuda.synchronize()
    cfg = load_model_config(overrides=over)
    r = Oracle(sd, cfg, dtype=torch.float32).reactive_replay(batch, 60, tf_cfg_name="teacher_forcing_training", eps=eps,
                                                              rollout_prior=rollout_prior, action_eps=act)
    buf = out["rollout_buffer"]
    assert (buf.valid.cpu() == r["valid"]).all() and (buf.override_masks.cpu() == r["override_masks"]).all()
    assert r["override_masks"][:, :, 10:].sum() == 0  # teacher_forcing_training: nothing is forced after the warm start
    assert np.abs(buf.latent_sample.cpu().numpy() - (r["prior_mean"] if rollout_prior else r["post_mean"]).numpy()
                  - eps * float(np.exp(-1.0))).max() <= 1e-6
    r64 = Oracle(sd, cfg, dtype=torch.float64).reactive_replay(batch, 60, tf_cfg_name="teacher_forcing_training", eps=eps,
                                                                rollout_prior=rollout_prior, action_eps=act)
    mem = _oracle_ensemble(lambda pb, e_, d_, a_, g_: Oracle(sd, cfg, dtype=torch.float32, gemm_order_seed=g_).reactive_replay(
        pb, 60, tf_cfg_name="teacher_forcing_training", eps=e_, rollout_prior=rollout_prior, action_eps=a_), batch, eps=eps, act=act, seed=3)
    _assert_closed_loop(buf.preds.cpu().numpy(), r, r64, f"training replay prior={rollout_prior} sampled={sampled_actions}", mem)
    assert np.abs(buf.action_log_probs.cpu().numpy() - r["action_log_probs"].numpy()).max() <= (2e-4 if sampled_actions else 1e-6)
    gv, gs = r["gt_valid"][:, 1:61].transpose(1, 2), r["gt_state"][:, 1:61].transpose(1, 2)
    rew, rv = TO.differentiable_reward(buf.valid.cpu(), buf.preds.cpu(), gv, gs, r["agent_size"], cfg["differentiable_reward"])
    st = TO.training_metric_states(buf.valid.cpu(), rv, rew, buf.override_masks.cpu(), r["agent_role"], r["dest_logits_raw"], r["goal_valid"],
                                   r["gt_dest"], r["post_mean"], r["post_log_std"], r["post_valid"], r["prior_mean"], r["prior_log_std"],
                                   r["prior_valid"], cfg["training_metrics"])
    want = TO.training_metric_compute(st, cfg["training_metrics"], "training")
    got = out["metrics_dict"]
    assert set(got) == set(want)
    for k in want:
        assert abs(got[k] - want[k]) <= 2e-5 * max(1.0, abs(want[k])), (k, got[k], want[k])
    assert out["loss"] == got["training/loss"]


def test_training_step_train_mode_masks_against_reference_golden():
    """The train-mode Bernoulli masks with explicit draws (VERDICT r02 missing #4): `pre_processing.input.dropout_p_history`
    (`sc_input.py:100-106`), `pre_processing.latent.dropout_p_history` (`sc_latent.py:171-173,216-218`) and `p_drop_hidden`
    (`waymo_motion.py:345-351`) through `WaymoMotion.training_step(history_keep=, hidden_drop=)`, against tests/golden/train_dropout.npz
    = the body of the reference's own training_step with `torch.bernoulli` / `torch.rand(1)` / the personality's rsample replaced by
    synth.make_train_draws (network in eval-mode arithmetic)."""
    from trafficbots_amd import synth
    from trafficbots_amd.config import load_model_config

    g, meta = load_golden("train_dropout")
    over = {"time_step_end": meta["time_step_end"], "n_joint_future": 1}
    over.update(meta["overrides"])
    cfg = load_model_config(overrides=over)
    sc, ov = meta["scene"], meta["overrides"]
    n_step = meta["time_step_end"] - cfg["time_step_sim_start"] + 1
    draws = synth.make_train_draws(meta["draws_seed"], meta["n_scene"], sc["n_agent"], sc["n_pl"], sc["n_tl"], n_step,
                                   ov["pre_processing.input.dropout_p_history"], ov["pre_processing.latent.dropout_p_history"], ov["p_drop_hidden"])
    batch = synth.make_val_batch(meta["base_seed"], meta["n_scene"], **sc)
    eps = torch.from_numpy(synth.make_latent_noise(meta["base_seed"] + 99, meta["n_scene"], sc["n_agent"])).cuda()
    wm = _engine(over, synth.make_state_dict(meta["weight_seed"]))
    keep = {k: torch.from_numpy(v) for k, v in draws.items() if k != "hidden_drop"}
    out = wm.training_step(batch, latent_eps=eps, history_keep=keep, hidden_drop=draws["hidden_drop"])
    torch.cuda.synchronize()
    buf = out["rollout_buffer"]
    assert (out["latent_post"].valid.cpu().numpy() == g["post_valid"]).all() and (out["latent_prior"].valid.cpu().numpy() == g["prior_valid"]).all()
    assert np.abs(out["latent_post"].mean.cpu().numpy() - g["post_mean"]).max() <= ONE_SHOT_TOL
    assert np.abs(out["latent_prior"].mean.cpu().numpy() - g["prior_mean"]).max() <= ONE_SHOT_TOL
    assert (buf.valid.cpu().numpy() == g["valid"]).all() and (buf.override_masks.cpu().numpy() == g["override_masks"]).all()
    d = np.abs(buf.preds.cpu().numpy() - g["preds"]) * g["valid"][..., None]
    assert d[..., :2].max() <= 1e-4, d[..., :2].max()  # (50 steps of a 10-agent scene)
    fh = g["final_hidden"].reshape(buf.final["final_hidden"].shape)
    assert np.abs(buf.final["final_hidden"].cpu().numpy() - fh).max() <= 1e-4
    ref = json.loads(bytes(g["metrics_json"]).decode())
    got = out["metrics_dict"]
    assert set(got) == set(ref) and all(abs(got[k] - ref[k]) <= 2e-4 * max(1.0, abs(ref[k])) for k in ref), (got, ref)
    assert np.allclose(out["train_states"].cpu().numpy(), g["train_states"], rtol=2e-4)
    # the masks matter, each of them: without the hidden drop / without the history masks the rollout is a different one
    plain = wm.training_step(batch, latent_eps=eps)["rollout_buffer"]
    nohid = wm.training_step(batch, latent_eps=eps, history_keep=keep)["rollout_buffer"]
    assert float((plain.preds - buf.preds).abs().max()) > 1e-2 and float((nohid.preds - buf.preds).abs().max()) > 1e-3
    # and drawn here when the configuration asks for them and none are given: reproducible from the generator
    a = wm.training_step(batch, latent_eps=eps, generator=torch.Generator(device="cuda").manual_seed(5))["rollout_buffer"].preds
    b = wm.training_step(batch, latent_eps=eps, generator=torch.Generator(device="cuda").manual_seed(5))["rollout_buffer"].preds
    c = wm.training_step(batch, latent_eps=eps, generator=torch.Generator(device="cuda").manual_seed(6))["rollout_buffer"].preds
    assert torch.equal(a, b) and not torch.equal(a, c)


def test_training_step_perturbed_latent_inputs_against_reference_golden():
    """`pre_processing.latent.perturb_input_to_latent` (round 5; `sc_latent.py:115-152`): in train mode both personality encoders see the
    episode re-centred in a frame drawn per scene.  Golden `train_perturb` = the body of the reference's training_step with its two
    `torch.rand` draws replaced by synth.make_latent_perturb: posterior / prior personalities, the replay they drive, the losses."""
    from trafficbots_amd import synth

    g, meta = load_golden("train_perturb")
    over = {"time_step_end": meta["time_step_end"], "n_joint_future": 1}
    over.update(meta["overrides"])
    sc = meta["scene"]
    batch = synth.make_val_batch(meta["base_seed"], meta["n_scene"], **sc)
    eps = torch.from_numpy(synth.make_latent_noise(meta["base_seed"] + 99, meta["n_scene"], sc["n_agent"])).cuda()
    lp = {k: torch.from_numpy(v).cuda() for k, v in synth.make_latent_perturb(meta["perturb_seed"], meta["n_scene"]).items()}
    wm = _engine(over, synth.make_state_dict(meta["weight_seed"]))
    out = wm.training_step(batch, latent_eps=eps, latent_perturb=lp)
    torch.cuda.synchronize()
    buf = out["rollout_buffer"]
    assert (out["latent_post"].valid.cpu().numpy() == g["post_valid"]).all() and (out["latent_prior"].valid.cpu().numpy() == g["prior_valid"]).all()
    e_post = float(np.abs(out["latent_post"].mean.cpu().numpy() - g["post_mean"]).max())
    e_prior = float(np.abs(out["latent_prior"].mean.cpu().numpy() - g["prior_mean"]).max())
    REPORT["train_perturb"] = {"post_mean": e_post, "prior_mean": e_prior}
    assert e_post <= ONE_SHOT_TOL and e_prior <= ONE_SHOT_TOL, (e_post, e_prior)
    assert (buf.valid.cpu().numpy() == g["valid"]).all() and (buf.override_masks.cpu().numpy() == g["override_masks"]).all()
    d = np.abs(buf.preds.cpu().numpy() - g["preds"]) * g["valid"][..., None]
    assert d[..., :2].max() <= 1e-4, d[..., :2].max()
    ref = json.loads(bytes(g["metrics_json"]).decode())
    got = out["metrics_dict"]
    assert set(got) == set(ref) and all(abs(got[k] - ref[k]) <= 2e-4 * max(1.0, abs(ref[k])) for k in ref), (got, ref)
    # the frame matters (the personalities of the unperturbed episode are different ones) and the zero frame is the identity
    wm0 = _engine({k: v for k, v in over.items() if "perturb" not in k}, synth.make_state_dict(meta["weight_seed"]))
    plain = wm0.training_step(batch, latent_eps=eps)
    assert float((plain["latent_post"].mean - out["latent_post"].mean).abs().max()) > 1e-3
    half = {"yaw": torch.full_like(lp["yaw"], 0.5), "pos": torch.full_like(lp["pos"], 0.5)}  # u = 0.5 -> yaw 0, position 0
    ident = wm.training_step(batch, latent_eps=eps, latent_perturb=half)
    assert torch.equal(ident["latent_post"].mean, plain["latent_post"].mean) and torch.equal(ident["rollout_buffer"].preds, plain["rollout_buffer"].preds)


@pytest.mark.parametrize("shape", [dict(n_agent=64, n_pl=256, n_tl=40), dict(n_agent=20, n_pl=33, n_tl=5, p_late_spawn=0.4, p_invalid_agent=0.2)])
def test_batched_warm_start_is_bit_identical(shape):
    """tb_rollout_io.warm_start_steps: the map / traffic-light attention halves of the teacher-forced steps run as one batched launch
    from the ground truth; every output must equal the step-by-step rollout bit for bit (late spawns included; histories with
    early exits are not eligible and never get the flag from the host mirror)."""
    from trafficbots_amd import synth
    from trafficbots_amd.config import load_model_config
    from trafficbots_amd.runtime import HipEngine, scene_from_batch

    cfg = load_model_config(overrides={"time_step_end": 30, "n_joint_future": 2})
    eng = HipEngine(cfg, "cuda:0")
    eng.load_state_dict(synth.make_state_dict(21))
    batch = synth.make_batch(4400, 6, **shape)
    s = scene_from_batch(batch, torch.device("cuda:0"))
    assert s["warm_ok"]
    enc = eng.encode_scene(s)
    feats = {"map_feature": enc["map_feature"], "map_feature_valid": enc["map_feature_valid"], "tl_feature": enc["tl_feature"]}
    a = shape["n_agent"]
    z = enc["latent_mean"].repeat_interleave(2, 0) + 0.3 * torch.from_numpy(synth.make_latent_noise(4401, 12, a)).cuda()
    dest = enc["dest_logits"].argmax(-1).to(torch.int32).repeat_interleave(2, 0)
    gv = s["agent_valid"].bool().any(1).to(torch.uint8).repeat_interleave(2, 0)
    outs = []
    for w in (-1, 10, 4):  # -1: explicitly off (0 would let the engine derive it from the scene)
        s2 = dict(s, warm_ok=False) if w < 0 else s
        o = eng.rollout(s2, feats, z, enc["latent_mean"], dest, gv, 2, 30, warm_start_steps=max(w, 0), tap_step=3)
        torch.cuda.synchronize()
        outs.append({k: v.clone() for k, v in o.items() if torch.is_tensor(v)})
    for other in outs[1:]:
        for k, v in outs[0].items():
            assert torch.equal(v, other[k]), k
    # a history with an early exit is flagged not eligible by the host
    assert not scene_from_batch(synth.make_batch(4402, 4, n_agent=16, n_pl=20, p_early_exit=0.9), torch.device("cuda:0"))["warm_ok"]


def test_row_tiles_of_an_instance_may_run_apart():
    """720 workgroups of 6 row tiles per instance (80 agents -> a_pad 96) do not fit the chip at once and 6 does not divide the 32 CUs of
    an XCD, so some instances have their tiles in different dispatch waves: a tile then runs its C(t) long after a sibling finished
    A(t+1).  The interaction K / V and the validity bytes the tiles exchange are double-buffered by step parity for exactly this case
    (tb_rollout.hpp); with one buffer this test differs by centimetres.  Results must equal those of the same scenes run 4 at a time."""
    from trafficbots_amd import synth

    a, p_, b, k, s_end = 80, 64, 40, 3, 20
    wm = _engine({"time_step_end": s_end, "n_joint_future": k}, synth.make_state_dict(11))
    batch = synth.make_batch(4200, b, n_agent=a, n_pl=p_, n_tl=8, p_late_spawn=0.2, p_invalid_agent=0.1)
    eps = synth.make_latent_noise(4201, b * k, a)
    big = wm.test_step(batch, latent_eps=torch.from_numpy(eps).cuda(), generator=torch.Generator(device="cuda").manual_seed(5))
    torch.cuda.synchronize()
    gs = big["goal_sample"]
    for b0 in (0, 12, 24, 28):  # (with this shape the straddling instances are 40..47 and 80..87: scenes 13..15 and 26..29)
        sub = {key: v[b0:b0 + 4] for key, v in batch.items()}
        e = eps.reshape(b, k, a, -1)[b0:b0 + 4].reshape(4 * k, a, -1)
        small = wm.test_step(sub, latent_eps=torch.from_numpy(e).cuda(), goal_sample=gs[b0:b0 + 4].transpose(1, 2).contiguous())
        torch.cuda.synchronize()
        assert torch.equal(small["rollout_buffer"].valid, big["rollout_buffer"].valid[b0:b0 + 4])
        assert torch.equal(small["rollout_buffer"].preds, big["rollout_buffer"].preds[b0:b0 + 4]), b0


def test_packed_h5_loader_drives_validation_and_test_steps(tmp_path):
    """SURVEY 8(f)-4 end to end: episodes written in the reference's packed-h5 format and read back by the native loader (decoded on
    the host into the C-ABI layout, uploaded from pinned buffers) give bit-identical validation_step / test_step results to the same
    episodes handed over as an in-memory batch of the reference's layout."""
    from conftest import require_h5

    require_h5()
    from trafficbots_amd import data_h5, synth

    scene = dict(n_agent=20, n_pl=48, n_tl=12, p_invalid_agent=0.2, p_late_spawn=0.3, p_future_spawn=0.4, p_future_exit=0.3, pos_range=60.0)
    episodes, attrs = synth.make_h5_episodes(9900, 5, **scene)
    data_h5.write_packed_h5(str(tmp_path / "validation.h5"), episodes, attrs)
    data_h5.write_packed_h5(str(tmp_path / "testing.h5"), [{k: v for k, v in e.items() if k.startswith(("history/", "map/"))} for e in episodes], attrs)
    dm = data_h5.DataH5womd(str(tmp_path), batch_size=2, n_agent=20, n_pl=48, n_tl_stop=12)
    for table in (dm.tensor_size_val, dm.tensor_size_test):
        for k in list(table):
            table[k] = episodes[0][k].shape  # (the synthetic agent_no_sim / tl_lane tensors are smaller than Waymo's)
    sd = synth.make_state_dict(9901)
    wm = _engine({"time_step_end": 90, "n_joint_future": 1}, sd)
    dm.setup("validate")
    n = 0
    for batch in dm.val_dataloader():
        idx = batch["episode_idx"].tolist()
        assert batch["packed/agent_pos"].is_pinned()
        mem = {k: np.stack([episodes[i][k] for i in idx]) for k in episodes[0]}
        a, b = wm.validation_step(batch), wm.validation_step(mem)
        for part in ("reactive_replay", "joint_future_pred"):
            assert torch.equal(a[part]["rollout_buffer"].preds, b[part]["rollout_buffer"].preds), part
            assert torch.equal(a[part]["rollout_buffer"].valid, b[part]["rollout_buffer"].valid), part
        assert torch.equal(a["reactive_replay"]["train_states"], b["reactive_replay"]["train_states"])
        assert torch.equal(a["latent_post"].mean, b["latent_post"].mean)
        n += len(idx)
    assert n == 5
    # a training file holds no "history/*" tensors: the scene part is the first 11 steps of the ground truth (n_lead in the reader)
    data_h5.write_packed_h5(str(tmp_path / "training.h5"), [{k: e[k] for k in dm.tensor_size_train} for e in episodes])
    for k in list(dm.tensor_size_train):
        dm.tensor_size_train[k] = episodes[0][k].shape
    dm.setup("fit")
    for batch in dm.train_dataloader():
        idx = batch["episode_idx"].tolist()  # random draws (DatasetTrain)
        mem = {k: np.stack([episodes[i][k] for i in idx]) for k in episodes[0]}
        eps = torch.from_numpy(synth.make_latent_noise(9904, len(idx), 20)).cuda()
        a, b = wm.training_step(batch, latent_eps=eps), wm.training_step(mem, latent_eps=eps)
        assert torch.equal(a["train_states"], b["train_states"]) and a["loss"] == b["loss"]
        assert torch.equal(a["rollout_buffer"].preds, b["rollout_buffer"].preds)
    wm3 = _engine({"time_step_end": 40, "n_joint_future": 3}, sd)
    dm.setup("test")
    for batch in dm.test_dataloader():
        idx = batch["episode_idx"].tolist()
        mem = {k: np.stack([episodes[i][k] for i in idx]) for k in episodes[0] if k.startswith(("history/", "map/"))}
        eps = torch.from_numpy(synth.make_latent_noise(9902, len(idx) * 3, 20)).cuda()
        gen = lambda: torch.Generator(device="cuda").manual_seed(9903)  # the K destination draws
        a, b = wm3.test_step(batch, latent_eps=eps, generator=gen()), wm3.test_step(mem, latent_eps=eps, generator=gen())
        assert torch.equal(a["goal_sample"], b["goal_sample"])
        assert torch.equal(a["rollout_buffer"].preds, b["rollout_buffer"].preds)


def test_empty_and_bad_inputs_fail_loudly():
    from trafficbots_amd import synth
    from trafficbots_amd.waymo_motion import WaymoMotion

    wm = WaymoMotion()
    batch = synth.make_batch(1, 1, n_agent=8, n_pl=16)
    with pytest.raises(RuntimeError):  # weights not loaded
        wm.test_step(batch)
    sd = synth.make_state_dict(7)
    bad = dict(sd)
    bad.pop("model.agent_temporal.rnn.weight_hh_l1")
    with pytest.raises(RuntimeError):
        wm.load_state_dict(bad)
    with pytest.raises(NotImplementedError):
        WaymoMotion(**{"model.tf_cfg.n_head": 8})


SUITE_CASES = ("small_k1", "masks_k3", "degenerate", "headline_2", "headline_k6", "headline_w_normal", "headline_w_sharp",
               "headline_w_ln_gamma", "headline_8", "stress_1", "stoch_actions", "val_small", "val_masks", "val_alt_losses", "val_irrelevant",
               "headline_w_ckpt")


def test_suite_level_closed_loop_parity():
    """VERDICT r03 task 1 (b): the per-case bounds are prediction limits of ONE further run and leave room for a path that sits at the
    edge of every one of them; over the suite that is not allowed (tools/ensemble.py::suite_rule): the number of cases in which the HIP
    path ends farther from the reference's fp64 twin than all 33 reference runs <= the 99 % binomial quantile for p = 1/34, and the
    geometric mean over the cases of (HIP's final distance to fp64) / (the median member's) <= 1.5.  Runs after the golden tests of this
    module (file order) and needs all of them."""
    from tools import ensemble

    missing = [c for c in SUITE_CASES if c not in SUITE]
    assert not missing, f"suite-level rule needs the golden tests of this module to have run: missing {missing}"
    r = ensemble.suite_rule({c: SUITE[c] for c in SUITE_CASES})
    REPORT["suite_level"] = r
    assert r["ok"], r


def test_zz_write_report():
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "parity_report.json"), "w") as f:
        json.dump(REPORT, f, indent=1, sort_keys=True)
    print(json.dumps(REPORT, indent=1, sort_keys=True))


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
@pytest.mark.parametrize("shape", [dict(n_scene=32, n_agent=64, n_pl=256, n_tl=40), dict(n_scene=3, n_agent=20, n_pl=33, n_tl=5, p_late_spawn=0.4,
                                                                                     p_invalid_agent=0.2)])
def test_helper_workgroups_do_not_change_results(shape, precision, monkeypatch):
    """Launches of at most 128 tiles run a second workgroup per tile on the idle CUs.  It computes what depends only on the previous
    launch's data -- the interaction K / V of layers 1, 2 from the stored x_mid (kv_helper_x), then W_hh h of the three GRU layers from
    the hidden state (gru_hh_helper) -- and hands both to the tile workgroups through L2 (tb_device_xdl.hpp).  A tile's result must
    not depend on who computed them: with the helpers off (TB_STEP_HELPERS=0) the rollout is BITWISE the same, and so it is when the
    helpers start ~100 us late in every launch (the hand-off under uneven load: the tile workgroups must wait for the K / V flags and
    fall back to their own W_hh h, never read early)."""
    from trafficbots_amd import synth

    shape = dict(shape)
    n_scene = shape.pop("n_scene")
    sd = synth.make_state_dict(7)
    batch = synth.make_batch(8100, n_scene, **shape)
    eps = torch.from_numpy(synth.make_latent_noise(3, n_scene, shape["n_agent"])).cuda()
    outs = {}
    for name, flag, delay in (("on", "1", "0"), ("off", "0", "0"), ("late", "1", "220000")):
        monkeypatch.setenv("TB_STEP_HELPERS", flag)
        monkeypatch.setenv("TB_DEBUG_HELPER_DELAY", delay)
        wm = _engine({"time_step_end": 60 if name != "late" else 24, "n_joint_future": 1, "operand_precision": precision}, sd)
        outs[name] = wm.test_step(batch, latent_eps=eps)["rollout_buffer"]
        wm.engine.check_status()  # (no hand-off time-out, no range overflow)
    torch.cuda.synchronize()
    a, b, c = outs["on"], outs["off"], outs["late"]
    assert torch.isfinite(a.preds).all()
    assert torch.equal(a.preds, b.preds), float((a.preds - b.preds).abs().max())
    assert torch.equal(a.valid, b.valid) and torch.equal(a.final["final_hidden"], b.final["final_hidden"])
    n = c.preds.shape[-2]
    assert torch.equal(a.preds[..., :n, :], c.preds), float((a.preds[..., :n, :] - c.preds).abs().max())


def test_helper_handoff_timeout_fails_loudly(monkeypatch):
    """a tile workgroup that gives up waiting for its helper must not continue silently on stale K / V.  With the
    helpers held back for ~0.5 s per launch (TB_DEBUG_HELPER_DELAY beyond the bounded spin of kv_wait_x) the rollout's trajectories
    come out NaN (the epilogue poisons the step) AND tb_check_status reports the time-out."""
    from trafficbots_amd import synth

    sd = synth.make_state_dict(7)
    batch = synth.make_batch(8100, 3, n_agent=20, n_pl=33, n_tl=5)
    eps = torch.from_numpy(synth.make_latent_noise(3, 3, 20)).cuda()
    monkeypatch.setenv("TB_STEP_HELPERS", "1")
    monkeypatch.setenv("TB_DEBUG_HELPER_DELAY", "1200000000")
    # (round 5: the interaction of the teacher-forced steps comes out of the batched warm start, where no hand-off exists; the two
    # steps of this case are such steps, so the batching is switched off to keep the helper -> tile hand-off under test)
    monkeypatch.setenv("TB_STEP_PRE_INTER", "0")
    wm = _engine({"time_step_end": 2, "n_joint_future": 1}, sd)
    wm.check_range = False
    buf = wm.test_step(batch, latent_eps=eps)["rollout_buffer"]
    torch.cuda.synchronize()
    assert torch.isnan(buf.preds).any(), "a timed-out hand-off left plausible numbers behind"
    with pytest.raises(RuntimeError, match="helper"):
        wm.engine.check_status()
    wm.engine.check_status()  # (sticky word cleared by the report)
    monkeypatch.setenv("TB_DEBUG_HELPER_DELAY", "0")
    wm = _engine({"time_step_end": 2, "n_joint_future": 1}, sd)
    assert torch.isfinite(wm.test_step(batch, latent_eps=eps)["rollout_buffer"].preds).all()


def test_training_step_perturbation_with_input_dropout_against_reference_golden():
    """ADVICE r05 (medium), pinned by the reference itself in round 6: `perturb_input_to_latent` TOGETHER with
    `pre_processing.input.dropout_p_history` (+ `p_drop_hidden`).  The perturbed personality encoders see the UN-dropped episode
    (`sc_latent.py:115-152` builds latent_post / latent_prior from sc/* under perturbation) while the model inputs are dropped
    (`sc_input.py:100-106`).  Golden `train_perturb_dropout` = the body of the reference's training_step with the Bernoulli masks, the
    two frame draws, the hidden-state draws and the personality's rsample replaced by stored draws."""
    from trafficbots_amd import synth
    from trafficbots_amd.config import load_model_config

    g, meta = load_golden("train_perturb_dropout")
    over = {"time_step_end": meta["time_step_end"], "n_joint_future": 1}
    over.update(meta["overrides"])
    cfg = load_model_config(overrides=over)
    sc, ov = meta["scene"], meta["overrides"]
    n_step = meta["time_step_end"] - cfg["time_step_sim_start"] + 1
    draws = synth.make_train_draws(meta["draws_seed"], meta["n_scene"], sc["n_agent"], sc["n_pl"], sc["n_tl"], n_step,
                                   ov["pre_processing.input.dropout_p_history"], ov["pre_processing.latent.dropout_p_history"], ov["p_drop_hidden"])
    assert (np.nonzero(draws["hidden_drop"])[0] == g["hidden_drop_steps"]).all() and not draws["input_agent"].all()
    batch = synth.make_val_batch(meta["base_seed"], meta["n_scene"], **sc)
    eps = torch.from_numpy(synth.make_latent_noise(meta["base_seed"] + 99, meta["n_scene"], sc["n_agent"])).cuda()
    lp = {k: torch.from_numpy(v).cuda() for k, v in synth.make_latent_perturb(meta["perturb_seed"], meta["n_scene"]).items()}
    keep = {k: torch.from_numpy(v) for k, v in draws.items() if k.startswith("input_")}
    wm = _engine(over, synth.make_state_dict(meta["weight_seed"]))
    out = wm.training_step(batch, latent_eps=eps, history_keep=keep, hidden_drop=draws["hidden_drop"], latent_perturb=lp)
    torch.cuda.synchronize()
    buf = out["rollout_buffer"]
    assert (out["latent_post"].valid.cpu().numpy() == g["post_valid"]).all() and (out["latent_prior"].valid.cpu().numpy() == g["prior_valid"]).all()
    e_post = float(np.abs(out["latent_post"].mean.cpu().numpy() - g["post_mean"]).max())
    e_prior = float(np.abs(out["latent_prior"].mean.cpu().numpy() - g["prior_mean"]).max())
    REPORT["train_perturb_dropout"] = {"post_mean": e_post, "prior_mean": e_prior}
    assert e_post <= ONE_SHOT_TOL and e_prior <= ONE_SHOT_TOL, (e_post, e_prior)
    assert np.abs(out["input_feature_dict"]["map_feature"].cpu().numpy() - g["map_feature"]).max() <= ONE_SHOT_TOL if "input_feature_dict" in out else True
    assert (buf.valid.cpu().numpy() == g["valid"]).all() and (buf.override_masks.cpu().numpy() == g["override_masks"]).all()
    d = np.abs(buf.preds.cpu().numpy() - g["preds"]) * g["valid"][..., None]
    assert d[..., :2].max() <= 1e-4, d[..., :2].max()
    ref = json.loads(bytes(g["metrics_json"]).decode())
    got = out["metrics_dict"]
    assert set(got) == set(ref) and all(abs(got[k] - ref[k]) <= 2e-4 * max(1.0, abs(ref[k])) for k in ref), (got, ref)
    # each ingredient matters: without the frame draws / without the input masks the personalities or the rollout are different ones
    no_pert = wm.training_step(batch, latent_eps=eps, history_keep=keep, hidden_drop=draws["hidden_drop"],
                               latent_perturb={k: torch.zeros_like(v) for k, v in lp.items()})
    no_keep = wm.training_step(batch, latent_eps=eps, hidden_drop=draws["hidden_drop"], latent_perturb=lp,
                               history_keep={k: torch.ones_like(v) for k, v in keep.items()})
    assert float((no_pert["latent_post"].mean - out["latent_post"].mean).abs().max()) > 1e-3
    assert float((no_keep["rollout_buffer"].preds - buf.preds).abs().max()) > 1e-3
