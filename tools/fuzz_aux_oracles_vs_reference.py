#!/usr/bin/env python3
"""Differential fuzz of the two auxiliary CPU oracles against the imported reference (runs here only): `WaymoPostProcessing`
(top-k / MTR NMS / MPA NMS / temperature, random sizes and thresholds; `aggr_thresh` is not a runnable configuration of the
reference, tools/gen_golden_post.py) and the thirteen ErrorMetrics / TrafficRuleMetrics sum states (random buffers).  The committed
goldens pin five + three configurations; this draws random ones.  usage: python tools/fuzz_aux_oracles_vs_reference.py [n] [seed]
-> profiles/r06_aux_oracles_vs_reference_fuzz.txt"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import ref_shim  # noqa: E402
from oracle.metrics_oracle import FIELDS, metric_partials  # noqa: E402
from oracle.post_processing_oracle import post_process  # noqa: E402
from trafficbots_amd import synth  # noqa: E402

ERR = ("err_counter", "err_pos_meter", "err_rot_deg", "err_spd_m_per_s")
RULE = ("counter_agent", "counter_veh", "outside_map", "collided", "run_road_edge", "run_red_light", "passive", "goal_reached", "dest_reached")


def main() -> int:
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 4242)
    ref_shim.install()
    from data_modules.waymo_post_processing import WaymoPostProcessing
    from models.metrics.logging import ErrorMetrics, TrafficRuleMetrics

    lines, bad = [], 0
    for i in range(n):
        # ---- post-processing
        k_pred = int(rng.integers(1, 7))
        n_pred = int(rng.choice([k_pred, k_pred + int(rng.integers(1, 20))]))
        thr = lambda: [float(x) for x in rng.uniform(0.3, 4.0, 3)]  # noqa: E731
        c = dict(k_pred=k_pred, score_temperature=float(rng.choice([0.0, 1.0, 1e2])), mpa=thr() if rng.random() < 0.5 else [],
                 mtr=thr() if rng.random() < 0.5 else [], aggr=[], n_iter_em=3, use_ade=bool(rng.random() < 0.5))
        b, a, s = int(rng.integers(1, 4)), int(rng.integers(1, 14)), int(rng.integers(2, 81))
        valid, scores, trajs, agent_type = synth.make_post_inputs(int(rng.integers(1, 2**30)), b, a, n_pred, s)
        pp = WaymoPostProcessing(c["k_pred"], c["score_temperature"], c["mpa"], c["mtr"], c["aggr"], c["n_iter_em"], c["use_ade"])
        with torch.no_grad():
            ref = pp(torch.from_numpy(valid), torch.from_numpy(scores.copy()), torch.from_numpy(trajs.copy()), torch.from_numpy(agent_type))
        out = post_process(valid, scores, trajs, agent_type, c["k_pred"], c["score_temperature"], c["mpa"], c["mtr"], c["aggr"], c["n_iter_em"], c["use_ade"])
        wt = ref["waymo_trajs"].numpy()
        match = np.all(wt[:, 0, :, :, None, :] == trajs[:, :, None, :, 0, :2], -1)
        ref_idx = match.argmax(-1)
        o_ref, o_got = np.argsort(ref_idx, -1), np.argsort(out["mode_idx"], -1)
        same_modes = bool((match.sum(-1) == 1).all()) and np.array_equal(np.take_along_axis(ref_idx, o_ref, -1), np.take_along_axis(out["mode_idx"], o_got, -1))
        e_s = float(np.abs(np.take_along_axis(ref["waymo_scores"].numpy(), o_ref, -1) - np.take_along_axis(out["waymo_scores"], o_got, -1)).max())
        ok_pp = same_modes and e_s < 1e-6 and np.array_equal(out["waymo_valid"], ref["waymo_valid"].numpy())
        # ---- metric states
        mb, ma, mk, ms, tf = int(rng.integers(1, 4)), int(rng.integers(1, 22)), int(rng.integers(1, 7)), int(rng.integers(2, 91)), bool(rng.random() < 0.5)
        d = {k_: torch.from_numpy(v) for k_, v in synth.make_metric_inputs(int(rng.integers(1, 2**30)), mb, ma, mk, ms).items()}
        em, rm = ErrorMetrics("x", tf), TrafficRuleMetrics("x", tf)
        em.update(d["pred_valid"], d["pred_states"], d["gt_valid"], d["gt_states"], d["override_masks"], d["agent_role"])
        rm.update(d["pred_valid"], d["override_masks"], d["outside_map"], d["collided"], d["run_road_edge"], d["run_red_light"], d["passive"],
                  d["goal_reached"], d["dest_reached"], d["agent_type"])
        want = np.array([float(getattr(em, k_)) for k_ in ERR] + [float(getattr(rm, k_)) for k_ in RULE])
        vio = {k_: d[k_] for k_ in ("outside_map", "collided", "run_road_edge", "run_red_light", "passive", "goal_reached", "dest_reached")}
        got_d = metric_partials(d["pred_valid"], d["pred_states"], d["override_masks"], vio, d["agent_type"], d["agent_role"], d["gt_valid"], d["gt_states"], tf)
        got = np.array([got_d[k_] for k_ in FIELDS])
        ok_m = bool(np.allclose(got, want, rtol=2e-6, atol=0))
        bad += int(not (ok_pp and ok_m))
        lines.append(f"case {i:3d} post: B={b} A={a:2d} NP={n_pred:2d} K={k_pred} S={s:2d} T={c['score_temperature']:g} mpa={'y' if c['mpa'] else 'n'} mtr={'y' if c['mtr'] else 'n'} "
                     f"ade={int(c['use_ade'])} -> {'ok' if ok_pp else 'DIFFERS'} (scores {e_s:.1e});  metrics: B={mb} A={ma:2d} K={mk} S={ms:2d} tf={int(tf)} -> "
                     f"{'ok' if ok_m else 'DIFFERS ' + str(np.abs(got - want).max())}")
        print(lines[-1], flush=True)
    head = [f"# tools/fuzz_aux_oracles_vs_reference.py: post-processing and metric-state oracles against the imported reference, {n} random configurations",
            f"# each; {bad} differ.  Modes selected equal (as sets per agent), scores < 1e-6, valid equal; the thirteen metric states rtol 2e-6."]
    open(os.path.join(ROOT, "profiles", "r06_aux_oracles_vs_reference_fuzz.txt"), "w").write("\n".join(head + lines) + "\n")
    print("\n".join(head))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
