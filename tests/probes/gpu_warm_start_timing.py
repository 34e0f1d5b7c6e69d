"""Development probe (GPU box): per-launch averages with and without the batched warm start (tb_rollout_io.warm_start_steps)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from trafficbots_amd import synth  # noqa: E402
from trafficbots_amd.config import load_model_config  # noqa: E402
from trafficbots_amd.runtime import HipEngine, scene_from_batch  # noqa: E402

cfg = load_model_config(overrides={"time_step_end": 90, "n_joint_future": 1})
eng = HipEngine(cfg, "cuda:0")
eng.load_state_dict(synth.make_state_dict(7))
scene = scene_from_batch(synth.make_batch(5000, 32, n_agent=64, n_pl=256, n_tl=40), torch.device("cuda:0"))
enc = eng.encode_scene(scene)
feats = {"map_feature": enc["map_feature"], "map_feature_valid": enc["map_feature_valid"], "tl_feature": enc["tl_feature"]}
z = enc["latent_mean"].clone()
dest = enc["dest_logits"].argmax(-1).to(torch.int32)
gv = scene["agent_valid"].bool().any(1).to(torch.uint8)
for warm in (False, True, False, True):
    s = dict(scene, warm_ok=warm)
    out = None
    for _ in range(5):
        out = eng.rollout(s, feats, z, enc["latent_mean"], dest, gv, 1, 90, out=out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(30):
        if i == 29:
            eng.set_timing(True)
        out = eng.rollout(s, feats, z, enc["latent_mean"], dest, gv, 1, 90, out=out)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 30
    tm = eng.get_timing()
    eng.set_timing(False)
    print(f"warm start batched={warm}: {dt * 1e3:.3f} ms per rollout; fused launches {tm['n_fused']} x {tm['fused_ms'] / tm['n_fused'] * 1e3:.2f} us, "
          f"other launches {tm['edge_ms']:.3f} ms, prologue {tm['prologue_ms']:.3f} ms")
