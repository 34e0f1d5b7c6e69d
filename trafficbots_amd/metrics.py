"""State holders with the names of the reference's torchmetrics classes (`src/models/metrics/logging.py`,
`src/models/metrics/training.py`).

The "sum" states themselves are produced on the GPU (`tb_metric_partials`, `tb_train_partials`) as packed float64 vectors; these
classes only accumulate such vectors over batches, SUM them over ranks (torchmetrics' `dist_reduce_fx="sum"`, one all-reduce
of the packed vector instead of one per state) and form the ratios of `compute()`.
"""
from __future__ import annotations

from typing import Dict, Sequence

import torch
from torch import Tensor

from .runtime import METRIC_FIELDS, TRAIN_FIELDS


class _PackedSumMetric:
    fields: Sequence[str] = ()

    def __init__(self, prefix: str) -> None:
        self.prefix = prefix
        self.states: Tensor = None   # this rank's accumulator
        self._synced: Tensor = None  # the all-reduced copy `compute()` reads after `sync()`; dropped by update() / reset()

    def update(self, packed: Tensor) -> None:
        """Add one batch's packed states (float64 vector in `fields` order)."""
        assert packed.shape == (len(self.fields),)
        self.states = packed.clone() if self.states is None else self.states + packed
        self._synced = None

    def reset(self) -> None:
        self.states = None
        self._synced = None

    def sync(self, device=None) -> None:
        """SUM over the default process group (no-op for a single process).  Like torchmetrics, the reduction works on a COPY:
        calling it twice, or `update()` after it, does not multiply the accumulator by the world size.  Every rank takes part in
        the collective, also one that received no batch (`PackedSceneLoader` does not pad ranks): it contributes zeros."""
        import torch.distributed as dist

        if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
            self._synced = None
            return
        if self.states is None:
            if device is None:
                device = "cpu" if dist.get_backend() == "gloo" or not torch.cuda.is_available() else torch.device("cuda", torch.cuda.current_device())
            red = torch.zeros(len(self.fields), dtype=torch.float64, device=device)
        else:
            red = self.states.clone()
        dist.all_reduce(red, op=dist.ReduceOp.SUM)
        self._synced = red

    def state_dict(self) -> Dict[str, float]:
        s = self._synced if self._synced is not None else self.states
        if s is None:
            s = torch.zeros(len(self.fields), dtype=torch.float64)
        return {k: float(s[i]) for i, k in enumerate(self.fields)}


class ErrorMetrics(_PackedSumMetric):
    """`logging.py:9-65`; states = the first four entries of `METRIC_FIELDS`."""

    fields = METRIC_FIELDS[:4]

    def compute(self) -> Dict[str, float]:
        s = self.state_dict()
        return {f"{self.prefix}/err/pos_meter": s["err_pos_meter"] / s["err_counter"],
                f"{self.prefix}/err/rot_deg": s["err_rot_deg"] / s["err_counter"],
                f"{self.prefix}/err/spd_m_per_s": s["err_spd_m_per_s"] / s["err_counter"]}


class TrafficRuleMetrics(_PackedSumMetric):
    """`logging.py:68-140`; states = the last nine entries of `METRIC_FIELDS`."""

    fields = METRIC_FIELDS[4:]

    def compute(self) -> Dict[str, float]:
        s = self.state_dict()
        out = {}
        for k, den in (("outside_map", "counter_agent"), ("collided", "counter_agent"), ("run_road_edge", "counter_veh"),
                       ("run_red_light", "counter_veh"), ("passive", "counter_veh"), ("goal_reached", "counter_agent"),
                       ("dest_reached", "counter_agent")):
            out[f"{self.prefix}/traffic_rule/{k}"] = s[k] / s[den]
        return out


class TrainingMetrics(_PackedSumMetric):
    """`training.py:10-158`; `cfg` = the `training_metrics` config group."""

    fields = TRAIN_FIELDS

    def __init__(self, prefix: str, **cfg) -> None:
        super().__init__(prefix)
        self.cfg = cfg

    def compute(self) -> Dict[str, float]:
        s, c = self.state_dict(), self.cfg
        out = {f"{self.prefix}/loss": 0.0}
        if c["w_vae_kl"] > 0:
            out[f"{self.prefix}/vae_kl"] = c["w_vae_kl"] * s["vae_kl"] / s["vae_kl_counter"]
            out[f"{self.prefix}/loss"] += out[f"{self.prefix}/vae_kl"]
        if c["w_diffbar_reward"] > 0:
            out[f"{self.prefix}/diffbar_reward"] = c["w_diffbar_reward"] * s["diffbar_reward"] / s["diffbar_reward_counter"]
            out[f"{self.prefix}/loss"] += out[f"{self.prefix}/diffbar_reward"]
        if c["w_goal"] > 0:
            out[f"{self.prefix}/goal_loss"] = c["w_goal"] * s["goal_loss"] / s["goal_counter"]
            out[f"{self.prefix}/loss"] += out[f"{self.prefix}/goal_loss"]
        return out
