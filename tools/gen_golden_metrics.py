"""Golden vectors for ErrorMetrics / TrafficRuleMetrics (the thirteen torchmetrics "sum" states a multi-GPU run all-reduces):
runs the imported reference's `update()` on seeded synthetic buffers and stores the accumulated states in
tests/golden/metrics.npz.  Build container only."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import ref_shim  # noqa: E402
from trafficbots_amd import synth  # noqa: E402

CASES = {"k1": dict(seed=9500, n_scene=3, n_agent=12, k=1, n_step=90, tf=False),
         "k6": dict(seed=9501, n_scene=2, n_agent=20, k=6, n_step=90, tf=False),
         "k3_tf": dict(seed=9502, n_scene=2, n_agent=9, k=3, n_step=40, tf=True)}
ERR = ("err_counter", "err_pos_meter", "err_rot_deg", "err_spd_m_per_s")
RULE = ("counter_agent", "counter_veh", "outside_map", "collided", "run_road_edge", "run_red_light", "passive", "goal_reached",
        "dest_reached")


def main():
    ref_shim.install()
    from models.metrics.logging import ErrorMetrics, TrafficRuleMetrics

    save = {}
    for name, c in CASES.items():
        d = {k: torch.from_numpy(v) for k, v in synth.make_metric_inputs(c["seed"], c["n_scene"], c["n_agent"], c["k"], c["n_step"]).items()}
        em, rm = ErrorMetrics("x", c["tf"]), TrafficRuleMetrics("x", c["tf"])
        em.update(d["pred_valid"], d["pred_states"], d["gt_valid"], d["gt_states"], d["override_masks"], d["agent_role"])
        rm.update(d["pred_valid"], d["override_masks"], d["outside_map"], d["collided"], d["run_road_edge"], d["run_red_light"],
                  d["passive"], d["goal_reached"], d["dest_reached"], d["agent_type"])
        vals = [float(getattr(em, k)) for k in ERR] + [float(getattr(rm, k)) for k in RULE]
        save[name] = np.array(vals, np.float64)
        print(name, dict(zip(ERR + RULE, vals)))
    save["meta_json"] = np.frombuffer(json.dumps({"cases": CASES, "fields": list(ERR + RULE)}).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "metrics.npz"), **save)


if __name__ == "__main__":
    main()
