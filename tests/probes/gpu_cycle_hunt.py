"""Development probe: reference cycles among the per-step objects of the harness steps (cyclic garbage keeps device memory alive until
the cyclic collector happens to run -- its thresholds count objects, not bytes)."""
import collections
import gc
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from trafficbots_amd import synth  # noqa: E402
from trafficbots_amd.waymo_motion import WaymoMotion  # noqa: E402

w = WaymoMotion(time_step_end=40, n_joint_future=3)
w.load_state_dict(synth.make_state_dict(7))
data = [synth.make_batch(6000 + i, 8, n_agent=20, n_pl=50, n_tl=12) for i in range(3)]
vdata = [synth.make_val_batch(6100 + i, 4, n_agent=20, n_pl=50, n_tl=12) for i in range(2)]
eps = torch.from_numpy(synth.make_latent_noise(9, 24, 20)).cuda()
veps = torch.from_numpy(synth.make_latent_noise(9, 12, 20)).cuda()


def hunt(name, fn):
    fn()
    fn()
    torch.cuda.synchronize()
    gc.collect()
    gc.disable()
    gc.set_debug(gc.DEBUG_SAVEALL)
    fn()
    fn()
    torch.cuda.synchronize()
    n = gc.collect()
    c = collections.Counter(type(o).__name__ for o in gc.garbage)
    tens = [o for o in gc.garbage if torch.is_tensor(o)]
    print(f"{name}: {n} unreachable objects after two calls; {len(tens)} tensors among them ({sum(t.numel() * t.element_size() for t in tens if t.is_cuda) / 2**20:.2f} MB of CUDA views); types: {c.most_common(8)}")
    if os.environ.get("SHOW"):
        for o in gc.garbage[:40]:
            if not torch.is_tensor(o):
                print("   ", type(o).__name__, repr(o)[:160])
    gc.set_debug(0)
    gc.garbage.clear()
    gc.enable()


hunt("test_step", lambda: w.test_step(data[0], latent_eps=eps))
hunt("validation_step", lambda: w.validation_step(vdata[0], latent_eps=veps))
hunt("pipeline x4", lambda: list(w.pipeline([data[0], data[1], data[2], data[0]], lanes=2, kwargs_fn=lambda i: dict(latent_eps=eps))))
hunt("prefetch x3", lambda: [w.test_step(sb, latent_eps=eps) for sb in w.prefetch([data[0], data[1], data[2]])])
