"""tests/golden/ensg/<name>.npz: rounding-noise ensembles of THE REFERENCE whose members are independent of the base run.

Every member is an fp32 run of the imported reference (`/root/reference/src` through tools/ref_shim.py) on
  * a channel-re-labelled copy of the weights (tools/channel_perm.py): every `F.linear`, LayerNorm and attention product sums in
    another order -- what the batch permutation of tools/ensemble.py alone leaves identical in all members, and
  * a permuted batch (agent slots, polylines, stop points; every other member at another batch size) as before.
Destinations are forced to the base run's.  The base runs themselves are NOT recomputed: `preds` / `preds_fp64` / `valid` come from the
committed golden tests/golden/<name>.npz, so this file only adds to it.  Stored:
  ensg_d32 [M,S]    per member, per step: max |member - base fp32| over xy of the entries valid in both;
  ensg_d64 [M+1,S]  per member (row 0 = the base fp32 run): max |member - fp64 twin|;
  equiv_fp64        max |channel-permuted fp64 run - fp64 twin| over the rollout (the re-labelling is the same function: ~1e-13 m);
  flips             valid-flag disagreements between members and the base run (expected 0).
Run here only (the reference does not travel):  python tools/gen_golden_ensg.py [names...]
"""
from __future__ import annotations

import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

import ensemble  # noqa: E402
import gen_golden  # noqa: E402
import gen_golden_val  # noqa: E402

OUT_DIR = os.path.join(ROOT, "tests", "golden", "ensg")
N_MEMBER = 32

ROLLOUT_CASES = ("small_k1", "masks_k3", "degenerate", "headline_2", "headline_k6", "stoch_actions", "action_override",
                 "headline_w_normal", "headline_w_sharp", "headline_w_ln_gamma", "headline_8", "stress_1", "headline_w_trained",
                 "headline_w_ckpt", "far_scene")
VAL_CASES = ("val_small", "val_masks", "val_alt_losses", "val_irrelevant", "val_trained")


def _one(name: str) -> None:
    g = np.load(os.path.join(gen_golden.GOLDEN_DIR, f"{name}.npz"))
    is_val = name in VAL_CASES
    case = (gen_golden_val.CASES if is_val else gen_golden.CASES)[name]
    step_axis = 2 if is_val else 3
    p32, v32, p64, v64 = g["preds"], g["valid"], g["preds_fp64"], g["valid_fp64"]

    def run(dtype, perturb, channel_seed):
        if is_val:
            return gen_golden_val.run_reference(case, dtype, perturb=None if perturb is None else perturb[0], channel_seed=channel_seed)
        return gen_golden.run_reference(case, dtype, force_goal_sample=g["goal_sample"], perturb=perturb, channel_seed=channel_seed)

    t0 = time.time()
    e64 = run(torch.float64, None, 424242)
    equiv = float((np.abs(e64["preds"] - p64) * (e64["valid"] & v64)[..., None]).max())
    assert np.array_equal(e64["valid"], v64) and equiv < 1e-9, f"{name}: channel re-labelling is not the same function ({equiv:.3e})"
    d32, d64 = [], [ensemble.spread_per_step(p32, p64, v32 & v64, step_axis)]
    flips = 0
    n_member = case.get("n_ensg", N_MEMBER)
    for i in range(n_member):
        # every other member at another batch size (rollout cases; the validation generator has no padding path)
        m = run(torch.float32, (1000 * case["base_seed"] + 500 + i, (i % 2) * (1 + i // 4) if case["scene"]["n_agent"] <= 64 else 0),
                channel_seed=77000 + 131 * i)
        flips += int((m["valid"] != v32).sum())
        d32.append(ensemble.spread_per_step(m["preds"], p32, m["valid"] & v32, step_axis))
        d64.append(ensemble.spread_per_step(m["preds"], p64, m["valid"] & v64, step_axis))
    d32, d64 = np.stack(d32).astype(np.float32), np.stack(d64).astype(np.float32)
    os.makedirs(OUT_DIR, exist_ok=True)
    path = os.path.join(OUT_DIR, f"{name}.npz")
    meta = {"n_member": n_member, "channel_seed": "77000 + 131 i", "perturb_seed": "1000 base_seed + 500 + i", "golden": f"{name}.npz"}
    np.savez_compressed(path, ensg_d32=d32, ensg_d64=d64, equiv_fp64=np.float64(equiv), flips=np.int64(flips),
                        meta_json=np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8))
    fin32, fin64 = np.maximum.accumulate(d32, 1)[:, -1], np.maximum.accumulate(d64, 1)[:, -1]
    print(f"[{name}] {n_member} members in {time.time() - t0:.0f} s: fp64 equivalence {equiv:.1e}; valid flips {flips}; final |m - fp32| "
          f"median {np.median(fin32):.3e} max {fin32.max():.3e} log-std {np.log(fin32).std(ddof=1):.2f}; final |m - fp64| median "
          f"{np.median(fin64[1:]):.3e} max {fin64[1:].max():.3e} log-std {np.log(fin64[1:]).std(ddof=1):.2f} (base {fin64[0]:.3e})", flush=True)


def main() -> None:
    torch.set_num_threads(os.cpu_count())
    names = sys.argv[1:] or list(ROLLOUT_CASES + VAL_CASES)
    for name in names:
        _one(name)


if __name__ == "__main__":
    main()
