"""Development probe (GPU box): when, on the GPU's clock, does the side-stream upload of the next batch start / finish relative to the
start of the current step's work on the main stream?"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from trafficbots_amd import staging, synth  # noqa: E402
from trafficbots_amd.waymo_motion import WaymoMotion  # noqa: E402

batches = [synth.make_batch(5000 + 37 * i, 32, n_agent=64, n_pl=256, n_tl=40) for i in range(4)]
wm = WaymoMotion(time_step_end=90, n_joint_future=1)
wm.load_state_dict(synth.make_state_dict(7))
ev = {}
orig_stage = staging.HostStager.stage


def stage(self, batch):
    e0 = torch.cuda.Event(enable_timing=True)
    e0.record()
    r = orig_stage(self, batch)
    e1 = torch.cuda.Event(enable_timing=True)
    e1.record()
    ev["copy"] = (e0, e1)
    return r


staging.HostStager.stage = stage
orig_enc = wm.engine.encode_scene


def enc(scene):
    r = orig_enc(scene)
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    ev["enc_end"] = e
    return r


wm.engine.encode_scene = enc
mode = os.environ.get("MODE", "prefetch")
stream = [batches[i % 4] for i in range(8)]
for i, sb in enumerate(wm.prefetch(stream, encode=mode != "stage_only")):
    e_main = torch.cuda.Event(enable_timing=True)
    e_main.record()
    ev.clear()
    wm.test_step(sb)
    e_end = torch.cuda.Event(enable_timing=True)
    e_end.record()
    torch.cuda.synchronize()
    if "copy" in ev and i >= 3:
        c0, c1 = ev["copy"]
        print(f"step {i}: main work {e_main.elapsed_time(e_end):.2f} ms | side: upload begins at {e_main.elapsed_time(c0):.2f}, ends at {e_main.elapsed_time(c1):.2f}"
              + (f", encoders end at {e_main.elapsed_time(ev['enc_end']):.2f}" if "enc_end" in ev else ""), flush=True)
