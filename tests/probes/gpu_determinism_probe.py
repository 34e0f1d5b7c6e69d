"""Development probe (GPU box): is test_step repeatable for a fixed generator seed -- XDL and exact-fp32 kernels, K = 1 / 2, normal and scaled weights?"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import golden_inputs, load_golden  # noqa: E402
from test_gpu_boundary import _scaled  # noqa: E402
from trafficbots_amd.waymo_motion import WaymoMotion  # noqa: E402

g, meta = load_golden("small_k1")
cfg, sd0, batch, eps = golden_inputs(meta)
for scale in (1.0, 3e5):
    sd = _scaled(sd0, scale) if scale != 1.0 else sd0
    for prec in ("fp32", "fp32_exact"):
        for k in (1, 2):
            wm = WaymoMotion(time_step_end=15, n_joint_future=k, operand_precision=prec)
            wm.load_state_dict(sd)
            wm.check_range = False
            outs = []
            for rep in range(3):
                o = wm.test_step(batch, generator=torch.Generator(device="cuda").manual_seed(77))
                torch.cuda.synchronize()
                outs.append((o["rollout_buffer"].preds.clone(), o["latent_mean"].clone(), o["rollout_buffer"].latent_sample.clone(), o["dest_logits"].clone()))
            eq = [[bool(torch.equal(outs[0][i], outs[r][i])) for i in range(4)] for r in (1, 2)]
            d = float((outs[0][0] - outs[1][0]).abs().max())
            print(f"scale {scale:g} {prec} K={k}: equal(preds, latent_mean, latent_sample, dest_logits) run1 {eq[0]} run2 {eq[1]} max|dpreds| {d:.3e} finite {bool(torch.isfinite(outs[0][0]).all())}")
