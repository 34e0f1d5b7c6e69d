"""Development probe (GPU box): wall time of WaymoMotion.test_step at the headline shape, split into host pre-processing (numpy batch ->
device layout), and the rest (encoders, K rollouts, rule checks / post-processing), for K = 1 and K = 6."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from trafficbots_amd import synth  # noqa: E402
from trafficbots_amd.waymo_motion import WaymoMotion  # noqa: E402

batch = synth.make_batch(5000, 32, n_agent=64, n_pl=256, n_tl=40)
for k in (1, 6):
    wm = WaymoMotion(time_step_end=90, n_joint_future=k)
    wm.load_state_dict(synth.make_state_dict(7))
    for _ in range(3):
        out = wm.test_step(batch)
    torch.cuda.synchronize()
    n = 10
    t0 = time.perf_counter()
    for _ in range(n):
        scene = wm.pre_processing(batch)
    torch.cuda.synchronize()
    t_pre = (time.perf_counter() - t0) / n
    t0 = time.perf_counter()
    for _ in range(n):
        out = wm.test_step(batch)
    torch.cuda.synchronize()
    t_all = (time.perf_counter() - t0) / n
    print(f"K={k}: test_step {t_all * 1e3:.2f} ms per 32 scenes ({32 / t_all:.0f} scenes/s), of which pre_processing (host -> device) {t_pre * 1e3:.2f} ms")
