"""Development probe (GPU box): ms per tb_encode_scene at the headline shape (and two others); prints a checksum of the outputs."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from trafficbots_amd import synth  # noqa: E402
from trafficbots_amd.config import load_model_config  # noqa: E402
from trafficbots_amd.runtime import HipEngine, scene_from_batch  # noqa: E402

dev = torch.device("cuda:0")
cfg = load_model_config()
sd = synth.make_state_dict(7)
for (b, a, p) in ((32, 64, 256), (4, 128, 1024)):
    batch = synth.make_batch(5000, b, n_agent=a, n_pl=p, n_tl=40, p_invalid_pl=0.2, p_invalid_node=0.3)
    eng = HipEngine(cfg, str(dev))
    eng.load_state_dict(sd)
    scene = scene_from_batch(batch, dev)
    enc = eng.encode_scene(scene)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        enc = eng.encode_scene(scene)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    chk = sum(float(v.double().abs().sum()) for k, v in enc.items() if torch.is_tensor(v) and v.is_floating_point() and torch.isfinite(v).all())
    print(f"B={b} A={a} P={p}: {dt * 1e3:.3f} ms per encode; checksum {chk:.10e}")
