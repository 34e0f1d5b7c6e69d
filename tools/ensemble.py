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


def prediction_bound(devs: np.ndarray, alpha: float = 1e-3) -> np.ndarray:
    """One-sided (1 - alpha) prediction bound, per simulation step, for the deviation of ONE MORE correct run, from the M member
    deviations `devs` [M, S] (per-step max-abs xy distance of each member to the same anchor).

    Why not simply the max over the members (VERDICT r02's first formulation): a further correct run exceeds the max of M exchangeable
    runs with probability 1 / (M + 1) -- with 16 members and ~15 closed-loop cases about one case per suite run fails although nothing
    is wrong (observed on the first GPU run of round 3: HIP at 1.007x / 1.2x the 16-member max on 2 of 15 cases, while its RANK among
    the members' distances to the fp64 truth was unremarkable: 15/17, 10/17, 9/17, 6/17, 3/17, 16/17, 12/17 ...).  The deviation of a
    chaotic rollout grows multiplicatively, so log(deviation) is close to normal across members; the bound is the standard prediction
    limit of a normal sample, exp(mean + t_{M-1, 1-alpha} * sqrt(1 + 1/M) * std) of the members' running-max deviations, made
    monotone in t.  The width comes from the measured spread of the reference's own arithmetic and the stated significance level
    (ALPHA below: chosen in round 4 before the HIP path was measured against the new ensembles; pinned since round 5)."""
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


# ----------------------------------------------------------------------------------------------------------------------------------
# THE acceptance rule of the closed-loop parity tests (round 4; VERDICT r03 task 1 / ADVICE r03).  One rule, used by
# tests/test_gpu_parity.py (goldens and oracle-checked cases), tests/probes and bench.py alike.  It was written before the HIP path
# was run against the ensembles it refers to and AMENDED ONCE after that first run (QUANT, history below: a gate that moved after a
# measurement, however small and however motivated); since round 5 its constants and code are pinned by hash in
# tests/test_parity_rule.py::test_the_rule_is_pinned and the v1 verdict stays visible as `ok_without_quantisation_term`.  The yardstick is an ensemble of INDEPENDENT correct fp32 runs of the reference arithmetic:
# members that re-order the sums inside every Linear / LayerNorm / attention product (tools/channel_perm.py for the imported
# reference -> tests/golden/ensg/*.npz; Oracle(gemm_order_seed=) for oracle-made ensembles) on top of the batch permutation.  With
# such members neither triangle term of round 3 is needed, and none is used.
#   per case, per simulation step t (running max over steps):
#     (a) |hip - fp64 twin|(t)  <= max(FLOOR, PB(members' + base run's distances to the fp64 twin)(t))
#     (b) |hip - base fp32|(t)  <= max(FLOOR, PB(members' distances to the base fp32 run)(t))
#     (c) |hip - base fp32|(t)  <= FLOOR for every t <= flat_until (north_star's flat bound where it is attainable)
#   with PB = prediction_bound(alpha = ALPHA) + QUANT and FLOOR = north_star's 1e-4 m.
#   QUANT = 2^-16 m = one fp32 ulp of a coordinate in [128, 256) m (the maps extend to +-150 m): a max-abs distance between two fp32
#   trajectories is only resolved to that.  History, kept because the rule was written before it was used and then amended: v1 (commit "Closed-loop parity
#   guard tightened") had no QUANT term.  Its first GPU run (profiles/r04_rule_calibration.txt) had the HIP path outside on 2 of 20
#   cases by 3 % and 5 % (headline_8: 2.877e-4 against 2.793e-4 at step 88; an oracle-made ensemble whose members all sit within 1 % of
#   9.6e-5 m at step 63, HIP at 1.05e-4 m -- one ulp away) while the suite-level rule, ranks and ratios were unremarkable (geometric
#   mean HIP / median member 0.98).  A leave-one-out test on the REFERENCE's own members (tools/rule_calibration.py: each of the 1008
#   member runs of the 16 goldens judged by the rule fitted to the other members of its case -- no HIP number involved) showed v1's
#   false-alarm rate at 6 of 1008 = 6x its nominal alpha, every excess between 1.01x and 1.24x: where the members' spread collapses
#   below the resolution of the coordinates the log-normal limit is narrower than one ulp.  With QUANT the same test gives 2 of 1008.
#   tests/test_parity_rule.py pins that calibration.
#   per suite (suite_rule): over all cases with a reference-made ensemble,
#     (S1) #cases in which HIP ends farther from the fp64 twin than EVERY member <= the 99 % binomial quantile for p = 1 / (M + 1);
#     (S2) geometric mean over the cases of  HIP's final distance to fp64 / the median member's  <= 1.5
#          (a 2x regression of the whole path turns the suite red even when every per-case bound still holds).
ALPHA = 1e-3
FLOOR = 1e-4
QUANT = 2.0 ** -16
SUITE_GEOMEAN_MAX = 1.5


def closed_loop_rule(d32: np.ndarray, d64: np.ndarray, ens_d32: np.ndarray, ens_d64: np.ndarray, n_flat: int = 0) -> dict:
    """d32 / d64 [S]: per-step max-abs xy distance of the run under test to the base fp32 run / the fp64 twin; ens_d32 [M,S] members vs
    base fp32; ens_d64 [M(+1),S] members (and the base run) vs the fp64 twin; n_flat = number of leading steps under the flat bound.
    Returns the verdict and everything that is reported next to it."""
    b32, b64 = prediction_bound(ens_d32, ALPHA) + QUANT, prediction_bound(ens_d64, ALPHA) + QUANT
    lim32, lim64 = np.maximum(FLOOR, b32), np.maximum(FLOOR, b64)
    r32, r64 = np.maximum.accumulate(d32.astype(np.float64)), np.maximum.accumulate(d64.astype(np.float64))
    fin32, fin64 = np.maximum.accumulate(ens_d32, axis=1)[:, -1].astype(np.float64), np.maximum.accumulate(ens_d64, axis=1)[:, -1].astype(np.float64)
    ok_a, ok_b = bool((r64 <= lim64).all()), bool((r32 <= lim32).all())
    # (reported, not asserted: the verdict of the rule as it was first written, without the quantisation term)
    ok_v1 = bool((r64 <= np.maximum(FLOOR, b64 - QUANT)).all() and (r32 <= np.maximum(FLOOR, b32 - QUANT)).all())
    ok_c = bool(n_flat == 0 or r32[:n_flat].max() <= FLOOR)
    tiny = 1e-12
    return {
        "ok": ok_a and ok_b and ok_c, "ok_vs_fp64": ok_a, "ok_vs_fp32": ok_b, "ok_flat": ok_c, "ok_without_quantisation_term": ok_v1 and ok_c,
        "final_vs_fp32": float(r32[-1]), "final_vs_fp64": float(r64[-1]),
        "bound_vs_fp32": float(lim32[-1]), "bound_vs_fp64": float(lim64[-1]),
        "members_median_vs_fp32": float(np.median(fin32)), "members_max_vs_fp32": float(fin32.max()),
        "members_median_vs_fp64": float(np.median(fin64)), "members_max_vs_fp64": float(fin64.max()),
        "ratio_to_median_vs_fp32": float(r32[-1] / max(np.median(fin32), tiny)), "ratio_to_median_vs_fp64": float(r64[-1] / max(np.median(fin64), tiny)),
        "rank_vs_fp32": f"{int((fin32 >= r32[-1]).sum())}/{fin32.size}", "rank_vs_fp64": f"{int((fin64 >= r64[-1]).sum())}/{fin64.size}",
        "beyond_all_members_vs_fp64": bool(r64[-1] > fin64.max()), "n_member": int(ens_d32.shape[0]),
        "first_step_outside_vs_fp64": int(np.nonzero(r64 > lim64)[0][0]) if not ok_a else None,
        "first_step_outside_vs_fp32": int(np.nonzero(r32 > lim32)[0][0]) if not ok_b else None,
        "per_step": {"hip_vs_fp32": [float(f"{x:.3e}") for x in r32], "bound_vs_fp32": [float(f"{x:.3e}") for x in lim32],
                     "hip_vs_fp64": [float(f"{x:.3e}") for x in r64], "bound_vs_fp64": [float(f"{x:.3e}") for x in lim64]},
    }


def suite_rule(records: dict) -> dict:
    """(S1) / (S2) over `records` = {case: closed_loop_rule(...) output} of the cases whose distances are above the quantisation floor
    (a case in which HIP and every member end within 2e-5 m of the truth carries no information about a ratio)."""
    from scipy import stats

    use = {k: r for k, r in records.items() if max(r["final_vs_fp64"], r["members_median_vs_fp64"]) > 2e-5}
    n = len(use)
    if n == 0:
        return {"ok": True, "n_case": 0}
    m = min(r["n_member"] for r in use.values()) + 1  # (+ the base run)
    n_beyond = sum(r["beyond_all_members_vs_fp64"] for r in use.values())
    q = int(stats.binom.ppf(0.99, n, 1.0 / (m + 1)))
    geo = float(np.exp(np.mean([np.log(max(r["ratio_to_median_vs_fp64"], 1e-3)) for r in use.values()])))
    geo32 = float(np.exp(np.mean([np.log(max(r["ratio_to_median_vs_fp32"], 1e-3)) for r in use.values()])))
    return {"ok": bool(n_beyond <= q and geo <= SUITE_GEOMEAN_MAX), "n_case": n, "n_beyond_all_members_vs_fp64": int(n_beyond),
            "binomial_99pct_quantile": q, "geomean_ratio_to_median_vs_fp64": geo, "geomean_max": SUITE_GEOMEAN_MAX,
            "geomean_ratio_to_median_vs_fp32_reported_only": geo32,
            "cases": {k: {"ratio_vs_fp64": r["ratio_to_median_vs_fp64"], "rank_vs_fp64": r["rank_vs_fp64"],
                          "ratio_vs_fp32": r["ratio_to_median_vs_fp32"], "rank_vs_fp32": r["rank_vs_fp32"]} for k, r in use.items()}}
