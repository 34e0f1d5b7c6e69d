"""Development probe (any box): wall time of one 90-step oracle replay (2 scenes x 40 agents) against the torch CPU thread count --
the host-side cost of the oracle ensembles in the GPU tests."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.trafficbots_oracle import Oracle  # noqa: E402
from trafficbots_amd import synth  # noqa: E402
from trafficbots_amd.config import load_model_config  # noqa: E402

over = {"time_step_end": 90, "n_joint_future": 1}
scene = dict(n_agent=40, n_pl=100, n_tl=40, p_invalid_agent=0.2, p_late_spawn=0.2, p_future_spawn=0.5, p_future_exit=0.3, pos_range=35.0, p_tl_valid=0.6)
sd = synth.make_state_dict(9700)
batch = synth.make_val_batch(9701, 2, **scene)
cfg = load_model_config(overrides=over)
print("cpus", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "torch threads (default)", torch.get_num_threads(), flush=True)
for n in (torch.get_num_threads(), 16, 8, 4, 2, 1):
    torch.set_num_threads(n)
    for kw in ({}, {"gemm_order_seed": 5}):
        t = time.time()
        with torch.no_grad():
            Oracle(sd, cfg, dtype=torch.float32, **kw).reactive_replay(batch, 90)
        print(f"threads {n:3d} {kw}: {time.time() - t:.1f} s", flush=True)
