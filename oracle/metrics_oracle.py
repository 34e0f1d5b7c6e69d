"""CPU restatement of the reference's ErrorMetrics / TrafficRuleMetrics state updates -- TEST INFRASTRUCTURE ONLY.

Follows `src/models/metrics/logging.py` of zhejz/TrafficBots: `ErrorMetrics.update` (:20-54) and
`TrafficRuleMetrics.update` (:86-129); `cast_rad` is `src/utils/transform_utils.py:9-11`.  The thirteen sums are the
torchmetrics states (`dist_reduce_fx="sum"`) a multi-GPU run all-reduces.  Pinned by tests/golden/metrics.npz (states the
imported reference accumulated on seeded synthetic buffers, tools/gen_golden_metrics.py).
Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg may import this module.
"""
from __future__ import annotations

import math
from typing import Dict

import torch
from torch import Tensor

FIELDS = ("err_counter", "err_pos_meter", "err_rot_deg", "err_spd_m_per_s", "counter_agent", "counter_veh", "outside_map",
          "collided", "run_road_edge", "run_red_light", "passive", "goal_reached", "dest_reached")


def metric_partials(pred_valid: Tensor, pred_states: Tensor, override_masks: Tensor, violations: Dict[str, Tensor], agent_type: Tensor,
                    agent_role: Tensor, gt_valid: Tensor | None, gt_states: Tensor | None, loss_for_teacher_forcing: bool = False) -> Dict[str, float]:
    """pred_valid / override_masks / violations[*] [B,A,K,S] bool, pred_states [B,A,K,S,4], agent_type / agent_role [B,A,3] bool,
    gt_valid [B,A,S] / gt_states [B,A,S,4] (None: the four error sums stay 0)."""
    out = {k: 0.0 for k in FIELDS}
    if gt_valid is not None:
        relevant = agent_role.any(-1)[:, :, None, None]
        pv = pred_valid & relevant
        if not loss_for_teacher_forcing:
            pv = pv & ~override_masks
        ev = gt_valid[:, :, None, :] & pv
        g = gt_states[:, :, None].expand_as(pred_states).masked_fill(~ev[..., None], 0.0)
        s = pred_states.masked_fill(~ev[..., None], 0.0)
        out["err_counter"] = float(ev.sum())
        out["err_pos_meter"] = float(torch.norm(g[..., :2] - s[..., :2], dim=-1).sum())
        d = g[..., 2] - s[..., 2]
        wrapped = (d + math.pi) % (2 * math.pi) - math.pi
        out["err_rot_deg"] = float(torch.rad2deg(wrapped).abs().sum())
        out["err_spd_m_per_s"] = float((g[..., 3] - s[..., 3]).abs().sum())
    if loss_for_teacher_forcing:
        av = pred_valid.any(-1)
        vio = violations
    else:
        keep = pred_valid & ~override_masks
        vio = {k: v & keep for k, v in violations.items()}
        av = keep.any(-1)
    out["counter_agent"] = float(av.sum())
    out["counter_veh"] = float((av & agent_type[:, :, 0:1]).sum())
    for k in ("outside_map", "collided", "run_road_edge", "run_red_light", "passive", "goal_reached", "dest_reached"):
        out[k] = float(vio[k].any(-1).sum())
    return out
