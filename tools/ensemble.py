"""Mathematically equivalent re-orderings of a synthetic batch, for MEASURING the rounding noise of a closed-loop rollout.

The policy is permutation-equivariant in the agent slots and invariant to the order of map polylines / traffic-light stop
points (attention + max pooling; the pose PE is absolute), and a scene does not see its batch neighbours.  Running the SAME
arithmetic on a permuted / padded batch and un-permuting the outputs therefore gives a result that differs from the base run
only through the order of fp32 summation (softmax sums, attention-weighted value sums, blocked GEMM edges).  An ensemble of
such runs is the measured distribution of "a correct fp32 implementation of the reference arithmetic"; the parity tests bound
the HIP path by that ensemble instead of a hand-picked multiple of one fp32-vs-fp64 run.

Test infrastructure only (used by tools/gen_golden*.py with the imported reference, and by tests/probes with the oracle).
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import numpy as np

# position of the permuted axis INSIDE one scene (batch axis removed)
_AGENT_T = ("valid", "pos", "z", "vel", "spd", "acc", "yaw_bbox", "yaw_rate")      # [T, A, ...]
_AGENT_S = ("type", "role", "size", "object_id", "dest", "goal", "cmd")           # [A, ...]
_TL = ("valid", "state", "pos", "dir")                                              # [T, n_tl, ...]
_MAP = ("valid", "type", "pos", "dir")                                              # [P, ...]


class Perm:
    """Per-scene permutations: new slot j holds old slot perm[b][j]."""

    def __init__(self, agent: np.ndarray, pl: np.ndarray, tl: np.ndarray, n_scene: int, n_pad: int):
        self.agent, self.pl, self.tl, self.n_scene, self.n_pad = agent, pl, tl, n_scene, n_pad
        self.agent_inv = np.argsort(agent, axis=1)
        self.pl_inv = np.argsort(pl, axis=1)

    # ---- inputs (old order -> new order)
    def agents_fwd(self, x: np.ndarray, k: int = 1, axis: int = 1) -> np.ndarray:
        """x [B*k, A, ...] (agent axis `axis`) in the old order -> new order; instance n belongs to scene n // k."""
        out = np.empty_like(x)
        for n in range(x.shape[0]):
            out[n] = np.take(x[n], self.agent[n // k], axis=axis - 1)
        return out

    def dest_fwd(self, dest: np.ndarray, k: int = 1) -> np.ndarray:
        """dest [B*k, A] polyline indices in the old numbering and old agent order -> new numbering, new agent order."""
        out = np.empty_like(dest)
        for n in range(dest.shape[0]):
            b = n // k
            d = dest[n][self.agent[b]]
            out[n] = np.where(d >= 0, self.pl_inv[b][np.clip(d, 0, None)], d)
        return out

    # ---- outputs (new order -> old order)
    def agents_back(self, x: np.ndarray, k: int = 1, axis: int = 1) -> np.ndarray:
        out = np.empty_like(x)
        for n in range(x.shape[0]):
            out[n] = np.take(x[n], self.agent_inv[n // k], axis=axis - 1)
        return out


def permute_batch(batch: Dict[str, np.ndarray], seed: int, agents: bool = True, polylines: bool = True, tl: bool = True,
                  pad: Optional[Dict[str, np.ndarray]] = None) -> Tuple[Dict[str, np.ndarray], Perm]:
    """Returns (permuted batch, Perm).  `pad`: further scenes (same keys / sizes) appended AFTER the permuted ones, so that
    the member runs at another batch size; their outputs are cut off by the caller (first Perm.n_scene scenes)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    b = batch["map/valid"].shape[0]
    n_agent = batch["history/agent/valid"].shape[2]
    n_pl = batch["map/valid"].shape[1]
    n_tl = batch["history/tl_stop/valid"].shape[2]
    ident = lambda n: np.tile(np.arange(n), (b, 1))  # noqa: E731
    pa = np.stack([rng.permutation(n_agent) for _ in range(b)]) if agents else ident(n_agent)
    pp = np.stack([rng.permutation(n_pl) for _ in range(b)]) if polylines else ident(n_pl)
    pt = np.stack([rng.permutation(n_tl) for _ in range(b)]) if tl else ident(n_tl)
    perm = Perm(pa, pp, pt, b, 0 if pad is None else pad["map/valid"].shape[0])
    out: Dict[str, np.ndarray] = {}
    for key, v in batch.items():
        head, _, leaf = key.rpartition("/")
        w = v.copy()
        if head in ("history/agent", "agent"):
            ax = 1 if leaf in _AGENT_T else 0 if leaf in _AGENT_S else None
            assert ax is not None, key
            for i in range(b):
                w[i] = np.take(v[i], pa[i], axis=ax)
            if leaf == "dest":
                for i in range(b):
                    w[i] = perm.pl_inv[i][w[i]]
        elif head in ("history/tl_stop", "tl_stop"):
            assert leaf in _TL, key
            for i in range(b):
                w[i] = np.take(v[i], pt[i], axis=1)
        elif head == "map" and leaf in _MAP:
            for i in range(b):
                w[i] = np.take(v[i], pp[i], axis=0)
        elif head == "map" and leaf == "boundary" or head == "history/agent_no_sim":
            pass
        else:
            raise KeyError(f"permute_batch: no rule for {key}")
        out[key] = np.concatenate([w, pad[key]], 0) if pad is not None else w
    return out, perm


def spread_per_step(member: np.ndarray, base: np.ndarray, valid: np.ndarray, step_axis: int) -> np.ndarray:
    """max-abs xy difference per simulation step over the entries valid in both; inputs [..., S, 4] / [..., S] on `step_axis`."""
    ax = tuple(i for i in range(member.ndim) if i != step_axis)
    d = np.abs(member.astype(np.float64) - base.astype(np.float64)) * valid[..., None]
    return d[..., :2].max(axis=ax)


def prediction_bound(devs: np.ndarray, alpha: float = 5e-4) -> np.ndarray:
    """One-sided (1 - alpha) prediction bound, per simulation step, for the deviation of ONE MORE correct run, from the M member
    deviations `devs` [M, S] (per-step max-abs xy distance of each member to the same anchor).

    Why not simply the max over the members (VERDICT r02's first formulation): a further correct run exceeds the max of M exchangeable
    runs with probability 1 / (M + 1) -- with 16 members and ~15 closed-loop cases about one case per suite run fails although nothing
    is wrong (observed on the first GPU run of round 3: HIP at 1.007x / 1.2x the 16-member max on 2 of 15 cases, while its RANK among
    the members' distances to the fp64 truth was unremarkable: 15/17, 10/17, 9/17, 6/17, 3/17, 16/17, 12/17 ...).  The deviation of a
    chaotic rollout grows multiplicatively, so log(deviation) is close to normal across members; the bound is the standard prediction
    limit of a normal sample, exp(mean + t_{M-1, 1-alpha} * sqrt(1 + 1/M) * std) of the members' running-max deviations, made
    monotone in t.  alpha = 5e-4 per case keeps the family-wise false-failure rate of the whole suite near 1 %.  No free multiplier:
    the width comes from the measured spread of the reference's own arithmetic and the stated significance level."""
    from scipy import stats

    m = devs.shape[0]
    # (deviations are quantised by the fp32 ulp of the coordinates, 7.6e-6 m at 100 m: a floor of 1e-5 m keeps the zero / one-ulp
    # mixture of the first steps from blowing up the log-variance; the assertions' own floor is north_star's 1e-4 m)
    em = np.maximum.accumulate(np.maximum(devs.astype(np.float64), 1e-5), axis=1)
    lg = np.log(em)
    mu, sd = lg.mean(0), lg.std(0, ddof=1) if m > 1 else np.zeros(lg.shape[1])
    k = float(stats.t.ppf(1.0 - alpha, m - 1)) * np.sqrt(1.0 + 1.0 / m) if m > 1 else 0.0
    return np.maximum.accumulate(np.exp(mu + k * sd))


def rank_among(devs: np.ndarray, value: float) -> str:
    """"r/M": how many members end at least as far from the anchor as `value` (reported next to every bound)."""
    return f"{int((np.maximum.accumulate(devs, axis=1)[:, -1] >= value).sum())}/{devs.shape[0]}"
