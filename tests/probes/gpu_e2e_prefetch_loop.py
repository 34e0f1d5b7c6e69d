"""Development probe (GPU box): the prefetch loop of tools/e2e_bench.py alone (K from the environment), for rocprofv3 timelines and A/B of
launch-shaping switches."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from trafficbots_amd import synth  # noqa: E402
from trafficbots_amd.waymo_motion import WaymoMotion  # noqa: E402

k = int(os.environ.get("K", "1"))
n = int(os.environ.get("N", "16"))
chk = os.environ.get("CHECK", "1") == "1"
pre = os.environ.get("PREFETCH", "1") == "1"
batches = [synth.make_batch(5000 + 37 * i, 32, n_agent=64, n_pl=256, n_tl=40) for i in range(4)]
wm = WaymoMotion(time_step_end=90, n_joint_future=k)
wm.load_state_dict(synth.make_state_dict(7))
wm.check_range = chk
for rep in range(int(os.environ.get("REPS", "3"))):
    stream = [batches[i % 4] for i in range(n + 2)]
    lanes = int(os.environ.get("LANES", "0"))
    if lanes:  # `wm.pipeline`: results in order, range-checked
        it = iter(wm.pipeline(stream, lanes=lanes))
        next(it)
        next(it)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in it:
            pass
    else:
        it = iter(wm.prefetch(stream)) if pre else iter(stream)
        wm.test_step(next(it))
        wm.test_step(next(it))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for sb in it:
            wm.test_step(sb)
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / n
    print(f"K={k} lanes={os.environ.get('LANES', 0)} prefetch={pre} check_range={chk} WARM={os.environ.get('TB_STEP_WARM')} HELPERS={os.environ.get('TB_STEP_HELPERS')}: "
          f"{t * 1e3:.2f} ms per batch = {32 * 90 * k / t:.0f} scene-steps/s", flush=True)
