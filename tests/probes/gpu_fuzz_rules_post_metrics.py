"""Development probe (GPU box): the three stand-alone kernels of SURVEY 8(f)-1 / (f)-2 -- tb_rule_checks, tb_post_process,
tb_metric_partials -- against their CPU oracles over randomly drawn shapes, masks and configurations (FUZZ_SEED selects the draw).
Rule flags and selected modes must be EQUAL, scores within 1e-6, counters equal, error sums within 1e-5 relative.
    python tests/probes/gpu_fuzz_rules_post_metrics.py [n_cases]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.metrics_oracle import metric_partials  # noqa: E402
from oracle.post_processing_oracle import post_process  # noqa: E402
from oracle.rule_checks_oracle import rule_checks  # noqa: E402
from trafficbots_amd import synth  # noqa: E402
from trafficbots_amd.config import load_model_config  # noqa: E402
from trafficbots_amd.post_processing import WaymoPostProcessing  # noqa: E402
from trafficbots_amd.runtime import METRIC_FIELDS, RULE_KEYS, HipEngine, scene_from_batch  # noqa: E402

torch.set_num_threads(min(8, torch.get_num_threads()))  # (the CPU oracles: small matrices, see tests/probes/oracle_thread_timing.py)
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(int(os.environ.get("FUZZ_SEED", "4242")))
eng = HipEngine(load_model_config())
flags = {f"enable_check_{c}": True for c in ("collided", "run_road_edge", "run_red_light", "passive")}
n_true = {k: 0 for k in RULE_KEYS}
for ci in range(n_cases):
    # ---- rule checks: random walks over a dense synthetic scene
    a, p, t = int(rng.choice([1, 2, 15, 17, 33, 64, 65])), int(rng.choice([1, 16, 33, 64, 130])), int(rng.choice([1, 8, 33, 40]))
    b, k, s_len = int(rng.integers(1, 4)), int(rng.integers(1, 4)), int(rng.choice([5, 30, 80]))
    batch = synth.make_batch(50000 + ci, b, n_agent=a, n_pl=p, n_tl=t, pos_range=float(rng.choice([15.0, 40.0])), p_tl_valid=0.8,
                             p_invalid_pl=float(rng.choice([0.0, 0.3])), p_invalid_node=float(rng.choice([0.0, 0.4])))
    start = np.repeat(np.concatenate([batch["history/agent/pos"][:, -1], batch["history/agent/yaw_bbox"][:, -1],
                                      batch["history/agent/spd"][:, -1]], -1), k, 0)  # [N,A,4]
    step = rng.normal(0, 0.6, (b * k, a, s_len, 2)) * (rng.random((b * k, a, 1, 1)) < 0.8)  # some agents stand still (passive check)
    state = np.zeros((b * k, a, s_len, 4), np.float32)
    state[..., :2] = start[:, :, None, :2] + np.cumsum(step, 2)
    state[..., 2] = start[:, :, None, 2] + np.cumsum(rng.normal(0, 0.1, (b * k, a, s_len)), 2)
    state[..., 3] = np.abs(start[:, :, None, 3] * (rng.random((b * k, a, 1)) < 0.7) + rng.normal(0, 0.3, (b * k, a, s_len)))
    # a third of the agents drive straight through a traffic-light stop point along its lane direction (red-light check)
    tl_pos, tl_dir = batch["history/tl_stop/pos"][:, 1], batch["history/tl_stop/dir"][:, 1]  # [B,T,2] (stop points do not move)
    for n in range(b * k):
        for i in range(a):
            if rng.random() < 0.33:
                j, t0, v = int(rng.integers(0, t)), int(rng.integers(0, s_len)), float(rng.uniform(0.3, 1.5))
                d = tl_dir[n // k, j] / max(float(np.linalg.norm(tl_dir[n // k, j])), 1e-6)
                state[n, i, :, :2] = tl_pos[n // k, j] + d * (np.arange(s_len)[:, None] - t0 - 0.37) * v
                state[n, i, :, 2] = np.arctan2(d[1], d[0])
                state[n, i, :, 3] = v * 10.0
    valid = rng.random((b * k, a, s_len)) < 0.85
    scene = scene_from_batch(batch, eng.device)
    got = eng.rule_checks(scene, torch.from_numpy(state).cuda(), torch.from_numpy(valid).cuda(), k, flags)
    tb = {key: torch.from_numpy(np.asarray(v)) for key, v in batch.items()}
    ref = rule_checks(torch.from_numpy(state), torch.from_numpy(valid), k, eng.cfg["time_step_sim_start"], tb["history/agent/type"], tb["history/agent/size"], tb["map/valid"],
                      tb["map/type"], tb["map/pos"], tb["map/dir"], tb["history/tl_stop/valid"], tb["history/tl_stop/pos"], tb["history/tl_stop/state"])
    bad = [key for key in RULE_KEYS if not np.array_equal(got[key].cpu().numpy().astype(bool), ref[key].numpy())]
    for key in bad:
        g_, r_ = got[key].cpu().numpy().astype(bool), ref[key].numpy()
        where = np.argwhere(g_ != r_)
        print(f"    {key}: {len(where)} mismatches of {int(r_.sum())} raised; first (instance, agent, step): {where[:4].tolist()}; "
              f"hip {g_[tuple(where[0])]} oracle {r_[tuple(where[0])]}; state there {state[tuple(where[0])].tolist()}, valid {valid[tuple(where[0])]}")
        n_, i_, _ = where[0]
        print(f"    agent type {batch['history/agent/type'][n_ // k, i_].tolist()} path {state[n_, i_, :, :2].tolist()} valid {valid[n_, i_].tolist()}")
        print(f"    tl pos {tl_pos[n_ // k].tolist()} dir {tl_dir[n_ // k].tolist()} valid {batch['history/tl_stop/valid'][n_ // k].tolist()} "
              f"state {batch['history/tl_stop/state'][n_ // k, -1].tolist()}")
    for key in RULE_KEYS:
        n_true[key] += int(ref[key].sum())
    # ---- post-processing
    n_pred, k_pred, n_step = int(rng.choice([1, 3, 6, 12, 32])), int(rng.choice([1, 3, 6])), int(rng.choice([16, 80]))
    pv, scores, trajs, agent_type = synth.make_post_inputs(60000 + ci, b, a, n_pred, n_step)
    cfgs = [dict(mpa=[], mtr=[], aggr=[]), dict(mpa=[2.5, 1.0, 2.0], mtr=[], aggr=[]), dict(mpa=[], mtr=[2.5, 1.0, 2.0], aggr=[]),
            dict(mpa=[1.0, 1.0, 1.0], mtr=[], aggr=[])]
    c = cfgs[int(rng.integers(0, len(cfgs)))]
    temp, use_ade = float(rng.choice([-1.0, 0.5, 2.0])), bool(rng.integers(0, 2))
    pp = WaymoPostProcessing(eng, k_pred, temp, c["mpa"], c["mtr"], c["aggr"], 3, use_ade)
    out = pp(torch.from_numpy(pv), torch.from_numpy(scores), torch.from_numpy(trajs), torch.from_numpy(agent_type))
    want = post_process(pv, scores, trajs, agent_type, k_pred, temp, c["mpa"], c["mtr"], c["aggr"], 3, use_ade)
    idx, ref_idx = out["mode_idx"].cpu().numpy().astype(np.int64), want["mode_idx"].astype(np.int64)
    o_ref, o_got = np.argsort(ref_idx, -1), np.argsort(idx, -1)
    if not np.array_equal(np.take_along_axis(ref_idx, o_ref, -1), np.take_along_axis(idx, o_got, -1)):
        bad.append("post: selected modes")
    e_sc = float(np.abs(np.take_along_axis(want["waymo_scores"], o_ref, -1) - np.take_along_axis(out["waymo_scores"].cpu().numpy(), o_got, -1)).max())
    if e_sc > 1e-6:
        bad.append(f"post: scores {e_sc:.1e}")
    if not np.array_equal(out["waymo_valid"].cpu().numpy(), want["waymo_valid"]):
        bad.append("post: valid")
    # ---- metric partials
    d = {key: torch.from_numpy(v) for key, v in synth.make_metric_inputs(70000 + ci, b, a, k, s_len).items()}
    vio = {key: d[key] for key in ("outside_map", "collided", "run_road_edge", "run_red_light", "passive", "goal_reached", "dest_reached")}
    tf = bool(rng.integers(0, 2))
    use_gt = bool(rng.integers(0, 4))
    gm = eng.metric_partials(d["pred_valid"], d["pred_states"], d["override_masks"], vio, d["agent_type"], d["agent_role"],
                             d["gt_valid"] if use_gt else None, d["gt_states"] if use_gt else None, tf).cpu().numpy()
    wm_ = metric_partials(d["pred_valid"], d["pred_states"], d["override_masks"], vio, d["agent_type"], d["agent_role"],
                          d["gt_valid"] if use_gt else None, d["gt_states"] if use_gt else None, tf)
    wv = np.array([wm_[f] for f in METRIC_FIELDS])
    counters = [0, 4, 5, 6, 7, 8, 9, 10, 11, 12]
    if not np.array_equal(gm[counters], wv[counters]):
        bad.append(f"metrics: counters {gm[counters]} vs {wv[counters]}")
    if not np.allclose(gm[1:4], wv[1:4], rtol=1e-5, atol=1e-9):
        bad.append(f"metrics: sums {gm[1:4]} vs {wv[1:4]}")
    print(f"case {ci:2d} A={a:2d} P={p:3d} T={t:2d} B={b} K={k} S={s_len:2d} | NP={n_pred:2d} k={k_pred} {c} T={temp} ade={use_ade} score err {e_sc:.1e}"
          f" | tf={tf} gt={use_gt}  {'ok' if not bad else 'FAIL: ' + '; '.join(bad)}", flush=True)
    if bad:
        sys.exit(1)
print(f"all {n_cases} cases ok; rule flags raised in the references: {n_true}")
