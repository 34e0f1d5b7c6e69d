"""Scene-parallel sharding and the one collective of the path.

Rollout instances never interact, so N GPUs each take a contiguous block of scenes (the layout of the
reference's DDP run, one DataLoader shard per rank, `src/run.py:51-53`) and the only exchange is a SUM
all-reduce of packed metric partials (the reference's torchmetrics states use `dist_reduce_fx="sum"`,
`src/models/metrics/logging.py:15-18`) plus a MAX all-reduce of the elapsed time.  `backend="nccl"` is RCCL
on ROCm; the CPU tests drive the same code over gloo.
"""
from __future__ import annotations

from typing import Dict, Tuple

import torch
from torch import Tensor

PARTIAL_FIELDS = ("valid_agent_steps", "sum_abs_xy", "outside_map_final", "dest_reached_final", "scene_steps")


def shard_range(n_scene_global: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of scenes owned by `rank`; blocks differ by at most one scene."""
    base, rem = divmod(n_scene_global, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def metric_partials(preds: Tensor, valid: Tensor, outside_map: Tensor, dest_reached: Tensor, n_scene: int, n_step: int) -> Tensor:
    """Pack this rank's partial sums into one float64 vector (PARTIAL_FIELDS order).
    preds [N,A,S,4], valid/outside_map/dest_reached [N,A,S]."""
    v = valid.bool()
    return torch.stack([
        v.sum().double(),
        (preds[..., :2].abs() * v.unsqueeze(-1)).sum().double(),
        outside_map[..., -1].sum().double(),
        dest_reached[..., -1].sum().double(),
        torch.tensor(float(n_scene * n_step), device=preds.device, dtype=torch.float64),
    ])


def all_reduce_partials(partial: Tensor, elapsed_s: float, fields=PARTIAL_FIELDS) -> Tuple[Dict[str, float], float]:
    """SUM the partials and MAX the elapsed time over the default process group (no-op when not initialised).  `fields` names the
    entries of `partial` (default: PARTIAL_FIELDS; bench.py appends the reference's thirteen metric states, runtime.METRIC_FIELDS)."""
    import torch.distributed as dist

    t = torch.tensor([elapsed_s], device=partial.device, dtype=torch.float64)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        if dist.get_backend() == "gloo" and partial.is_cuda:  # dry runs of the N > 1 flow without RCCL: reduce through host memory
            hp, ht = partial.cpu(), t.cpu()
            dist.all_reduce(hp, op=dist.ReduceOp.SUM)
            dist.all_reduce(ht, op=dist.ReduceOp.MAX)
            partial.copy_(hp)
            t.copy_(ht)
        else:
            dist.all_reduce(partial, op=dist.ReduceOp.SUM)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return {k: float(partial[i]) for i, k in enumerate(fields)}, float(t.item())
