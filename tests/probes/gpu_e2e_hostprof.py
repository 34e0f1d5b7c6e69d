"""Development probe (GPU box): where the HOST time of WaymoMotion.test_step goes (cProfile, fresh batch per call)."""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from trafficbots_amd import synth  # noqa: E402
from trafficbots_amd.waymo_motion import WaymoMotion  # noqa: E402

k = int(os.environ.get("K", "1"))
batches = [synth.make_batch(5000 + 37 * i, 32, n_agent=64, n_pl=256, n_tl=40) for i in range(4)]
wm = WaymoMotion(time_step_end=90, n_joint_future=k)
wm.load_state_dict(synth.make_state_dict(7))
wm.check_range = False
for i in range(6):
    wm.test_step(batches[i % 4])
torch.cuda.synchronize()
# wall clock of the C call alone
lib = wm.engine.lib
orig = lib.tb_rollout
acc = []


def timed(*a):
    t0 = time.perf_counter()
    r = orig(*a)
    acc.append(time.perf_counter() - t0)
    return r


lib.tb_rollout = timed
for i in range(8):
    wm.test_step(batches[i % 4])
    torch.cuda.synchronize()
print("tb_rollout host wall ms (GPU idle at call):", [f"{x * 1e3:.2f}" for x in acc])
acc.clear()
for i in range(8):
    wm.test_step(batches[i % 4])
torch.cuda.synchronize()
print("tb_rollout host wall ms (back to back):   ", [f"{x * 1e3:.2f}" for x in acc])
lib.tb_rollout = orig
print("graph stats", wm.engine.graph_stats())
pr = cProfile.Profile()
pr.enable()
for i in range(8):
    wm.test_step(batches[i % 4])
    torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
