"""Leave-one-out calibration of the closed-loop acceptance rule on the REFERENCE's own runs (no HIP number involved).

Every member run of every reference-made ensemble (tests/golden/ensg/*.npz: 16 goldens x (32 members vs the base fp32 run + 33 runs vs
the fp64 twin)) is judged by tools/ensemble.py's per-step limit fitted to the OTHER members of its case.  A member of the ensemble is by
construction a correct fp32 run of the reference arithmetic, so the fraction flagged is the rule's false-alarm rate; nominal = ALPHA.
Usage: python tools/rule_calibration.py  > profiles/r04_rule_calibration.txt   (CPU, seconds)
"""
from __future__ import annotations

import glob
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import ensemble  # noqa: E402


def leave_one_out(quant: float):
    tot, flagged = 0, []
    for f in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "ensg", "*.npz"))):
        e, name = np.load(f), os.path.basename(f)[:-4]
        for key in ("ensg_d32", "ensg_d64"):
            d = e[key]
            for i in range(d.shape[0]):
                lim = np.maximum(ensemble.FLOOR, ensemble.prediction_bound(np.delete(d, i, 0), ensemble.ALPHA) + quant)
                r = np.maximum.accumulate(d[i].astype(np.float64))
                if (r > lim).any():
                    flagged.append((name, key, i, int(np.nonzero(r > lim)[0][0]) + 1, float((r / lim).max())))
            tot += d.shape[0]
    return tot, flagged


if __name__ == "__main__":
    for label, q in (("v1 (no quantisation term)", 0.0), ("v2 (QUANT = 2^-16 m, the rule in force)", ensemble.QUANT)):
        tot, fl = leave_one_out(q)
        print(f"{label}: {len(fl)} of {tot} reference member runs flagged by the rule fitted to the other members of their case "
              f"(rate {len(fl) / tot:.4f}, nominal alpha {ensemble.ALPHA})")
        for name, key, i, step, ratio in fl:
            print(f"    {name:22s} {key} member {i:2d}: first outside at step {step:3d}, worst ratio to the limit {ratio:.3f}")
