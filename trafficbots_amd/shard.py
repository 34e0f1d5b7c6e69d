"""Scene-parallel sharding and the one collective of the path.

Rollout instances never interact, so N GPUs each take a contiguous block of scenes (the layout of the
reference's DDP run, one DataLoader shard per rank, `src/run.py:51-53`) and the only exchange is ONE SUM
all-reduce per pass of a packed vector: the metric partials (the reference's torchmetrics states use `dist_reduce_fx="sum"`,
`src/models/metrics/logging.py:15-18`) followed by one slot per rank for the elapsed time (max taken locally).  `backend="nccl"` is RCCL
on ROCm; the CPU tests drive the same code over gloo.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch
from torch import Tensor

PARTIAL_FIELDS = ("valid_agent_steps", "sum_abs_xy", "outside_map_final", "dest_reached_final", "scene_steps")

# collectives this process has issued through all_reduce_partials (bench.py reports the count per pass: it must be 1)
N_COLLECTIVES = 0


def shard_range(n_scene_global: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of scenes owned by `rank`; blocks differ by at most one scene."""
    base, rem = divmod(n_scene_global, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def bind_to_gpu_numa_node(device_index: int) -> Dict[str, object]:
    """Pin the calling (launching) thread to the CPUs of the NUMA node the GPU hangs off -- 8 ranks x ~90 launches per 8 ms share the
    host; a launching thread that migrates across sockets pays for it in launch latency.  Looks the node up through sysfs from the
    device's PCI address; a container whose cpuset does not intersect that node (or a host without sysfs entries) is left as it is.
    Returns what was found, for the bench line."""
    import os

    info: Dict[str, object] = {"node": None, "bound": False}
    try:
        pr = torch.cuda.get_device_properties(device_index)
        addr = f"{getattr(pr, 'pci_domain_id', 0):04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
        with open(f"/sys/bus/pci/devices/{addr}/numa_node") as f:
            node = int(f.read().strip())
        info["pci"] = addr
        info["node"] = node
        if node < 0:
            return info
        cpus = set()
        with open(f"/sys/devices/system/node/node{node}/cpulist") as f:
            for part in f.read().strip().split(","):
                lo, _, hi = part.partition("-")
                cpus.update(range(int(lo), int(hi or lo) + 1))
        allowed = os.sched_getaffinity(0)
        both = cpus & allowed
        info["cpus_allowed"], info["cpus_on_node"] = len(allowed), len(both)
        if both and both != allowed:
            os.sched_setaffinity(0, both)
            info["bound"] = True
        elif both:
            info["bound"] = True  # (already inside the node)
    except Exception as e:  # noqa: BLE001  (best effort: never take a bench down over affinity)
        info["error"] = f"{type(e).__name__}: {e}"
    return info


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


def all_reduce_partials(partial: Tensor, elapsed_s: float, fields=PARTIAL_FIELDS, per_rank: Optional[Dict[str, float]] = None):
    """SUM the partials and MAX the elapsed time over the default process group with ONE collective (no-op when the group is not
    initialised): the vector that travels is [partials..., onehot_rank(elapsed)] -- rank r writes its elapsed time into slot
    r of a world-size tail of zeros, so that a single SUM all-reduce delivers every rank's time and the maximum is taken
    locally (a SUM cannot take a max; a MAX cannot sum; world_size extra doubles cost nothing next to a second collective).
    `fields` names the entries of `partial` (default: PARTIAL_FIELDS; bench.py appends the reference's thirteen metric states,
    runtime.METRIC_FIELDS).
    `per_rank` ({name: this rank's number}, same names on every rank): further one-hot tails of the SAME vector (device ordinal, host
    CPU time of the launching thread ...); then a third value is returned, {name: [value of rank 0, rank 1, ...]} with "elapsed_s"
    added -- what `bench.py --gpus N` prints as `ranks` (ranks_seen = the slots that arrived non-empty)."""
    import torch.distributed as dist

    global N_COLLECTIVES
    extra = list(per_rank.items()) if per_rank else []
    import os

    forced = os.environ.get("TB_BENCH_FORCE_DIST") == "1"  # (world size 1 through the real backend: a 1-GPU box's RCCL smoke test)
    if not (dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or forced)):
        res = {k: float(partial[i]) for i, k in enumerate(fields)}, float(elapsed_s)
        return res + ({"elapsed_s": [float(elapsed_s)], **{k: [float(v)] for k, v in extra}},) if per_rank is not None else res
    world, rank = dist.get_world_size(), dist.get_rank()
    n = partial.numel()
    packed = torch.zeros(n + world * (1 + len(extra)), device=partial.device, dtype=torch.float64)
    packed[:n] = partial.to(torch.float64)
    packed[n + rank] = float(elapsed_s)
    for j, (_, v) in enumerate(extra):
        packed[n + world * (1 + j) + rank] = float(v)
    N_COLLECTIVES += 1
    if dist.get_backend() == "gloo" and packed.is_cuda:  # dry runs of the N > 1 flow without RCCL: reduce through host memory
        hp = packed.cpu()
        dist.all_reduce(hp, op=dist.ReduceOp.SUM)
        packed = hp
    else:
        dist.all_reduce(packed, op=dist.ReduceOp.SUM)
    host = packed.cpu()
    partial.copy_(host[:n].to(partial.dtype))
    res = {k: float(host[i]) for i, k in enumerate(fields)}, float(host[n : n + world].max())
    if per_rank is None:
        return res
    ranks = {"elapsed_s": host[n : n + world].tolist()}
    for j, (k, _) in enumerate(extra):
        ranks[k] = host[n + world * (1 + j) : n + world * (2 + j)].tolist()
    return res + (ranks,)
