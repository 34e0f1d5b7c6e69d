"""Development probe (GPU box): host-side timeline of the prefetch loop -- when, after a step begins, tb_rollout is called / returns,
the next batch's staging starts / ends, and the range check starts / ends."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from trafficbots_amd import staging, synth  # noqa: E402
from trafficbots_amd.waymo_motion import WaymoMotion  # noqa: E402

k = int(os.environ.get("K", "1"))
batches = [synth.make_batch(5000 + 37 * i, 32, n_agent=64, n_pl=256, n_tl=40) for i in range(4)]
wm = WaymoMotion(time_step_end=90, n_joint_future=k)
wm.load_state_dict(synth.make_state_dict(7))
marks = []
t_step = [0.0]


def mark(name):
    marks.append((name, (time.perf_counter() - t_step[0]) * 1e3))


def wrap(obj, attr, name):
    orig = getattr(obj, attr)

    def f(*a, **kw):
        mark(name + " >")
        r = orig(*a, **kw)
        mark(name + " <")
        return r

    setattr(obj, attr, f)


wrap(wm.engine.lib, "tb_rollout", "tb_rollout")
wrap(wm.engine.lib, "tb_encode_scene", "tb_encode_scene")
wrap(wm.engine.lib, "tb_dest_sample", "tb_dest_sample")
wrap(wm.engine.lib, "tb_post_process", "tb_post_process")
wrap(wm.engine.lib, "tb_check_status", "tb_check_status")
wrap(staging.BatchPrefetcher, "advance", "advance")
wrap(staging.HostStager, "stage", "stage")
stream = [batches[i % 4] for i in range(10)]
for i, sb in enumerate(wm.prefetch(stream)):
    marks.clear()
    t_step[0] = time.perf_counter()
    wm.test_step(sb)
    mark("test_step returns")
    if i >= 7:
        print(" | ".join(f"{n} {t:.2f}" for n, t in marks), flush=True)
