"""Per-pass durations of consecutive rollouts (HIP events between passes) -- shows the clock / queue ramp after an idle GPU."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
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
out = eng.rollout(scene, feats, z, enc["latent_mean"], dest, gv, 1, 90)
torch.cuda.synchronize()
if os.environ.get("PRIME"):
    x = torch.zeros(1 << 20, device="cuda:0")
    t0 = time.perf_counter()
    for _ in range(int(os.environ["PRIME"])):
        x.add_(1.0)
    torch.cuda.synchronize()
    print(f"primed with {os.environ['PRIME']} tiny launches in {(time.perf_counter() - t0) * 1e3:.1f} ms")
for idle, timing_last in ((0.0, False), (0.5, False), (0.0, True), (0.0, True)):
    time.sleep(idle)
    n = 20
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    host = []
    ev[0].record()
    t0 = time.perf_counter()
    for i in range(n):
        if timing_last and i == n - 1:
            eng.set_timing(True)
        out = eng.rollout(scene, feats, z, enc["latent_mean"], dest, gv, 1, 90, out=out)
        ev[i + 1].record()
        host.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    gpu = [ev[i].elapsed_time(ev[i + 1]) for i in range(n)]
    if timing_last:
        print("   timing:", eng.get_timing())
        eng.set_timing(False)
    print(f"idle {idle}s timing_last={timing_last}: wall/pass {wall / n * 1e3:.3f} ms; gpu per pass (ms):", " ".join(f"{g:.2f}" for g in gpu))
    print("   host enqueue done at (ms):", " ".join(f"{h * 1e3:.1f}" for h in host[:12]))
