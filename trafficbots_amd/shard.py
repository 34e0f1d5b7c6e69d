"""Scene-parallel sharding and the one collective of the path.

Rollout instances never interact, so N GPUs each take a contiguous block of scenes (the layout of the
reference's DDP run, one DataLoader shard per rank, `src/run.py:51-53`) and the only exchange is ONE SUM
all-reduce per pass of a packed vector: the metric partials (the reference's torchmetrics states use `dist_reduce_fx="sum"`,
`src/models/metrics/logging.py:15-18`) followed by one slot per rank for the elapsed time (max taken locally).  `backend="nccl"` is RCCL
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
    """SUM the partials and MAX the elapsed time over the default process group with ONE collective (no-op when the group is not
    initialised): the vector that travels is [partials..., onehot_rank(elapsed)] -- rank r writes its elapsed time into slot
    r of a world-size tail of zeros, so that a single SUM all-reduce delivers every rank's time and the maximum is taken
    locally (a SUM cannot take a max; a MAX cannot sum; world_size extra doubles cost nothing next to a second collective).
    `fields` names the entries of `partial` (default: PARTIAL_FIELDS; bench.py appends the reference's thirteen metric states,
    runtime.METRIC_FIELDS)."""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return {k: float(partial[i]) for i, k in enumerate(fields)}, float(elapsed_s)
    world, rank = dist.get_world_size(), dist.get_rank()
    n = partial.numel()
    packed = torch.zeros(n + world, device=partial.device, dtype=torch.float64)
    packed[:n] = partial.to(torch.float64)
    packed[n + rank] = float(elapsed_s)
    if dist.get_backend() == "gloo" and packed.is_cuda:  # dry runs of the N > 1 flow without RCCL: reduce through host memory
        hp = packed.cpu()
        dist.all_reduce(hp, op=dist.ReduceOp.SUM)
        packed = hp
    else:
        dist.all_reduce(packed, op=dist.ReduceOp.SUM)
    host = packed.cpu()
    partial.copy_(host[:n].to(partial.dtype))
    return {k: float(host[i]) for i, k in enumerate(fields)}, float(host[n:].max())
