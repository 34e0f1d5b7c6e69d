"""Development probe (GPU box): rollout throughput at the other BASELINE.json configurations on one GPU -- K = 6 futures
(config 4) and the A = 128 / P = 1024 / 160-future-step stress shape (config 5's per-GPU share) -- for both operand precisions.
Prints one JSON line per case.  Not the headline metric (bench.py), not part of the test-suite."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from trafficbots_amd import synth  # noqa: E402
from trafficbots_amd.config import load_model_config  # noqa: E402
from trafficbots_amd.runtime import HipEngine, scene_from_batch  # noqa: E402

CASES = [
    dict(name="headline B=32 K=1 A=64 P=256 S=90", b=32, k=1, a=64, p=256, step_end=90),
    dict(name="config 4: B=32 K=6 A=64 P=256 S=90", b=32, k=6, a=64, p=256, step_end=90),
    dict(name="config 5 share: B=32 K=1 A=128 P=1024 S=170", b=32, k=1, a=128, p=1024, step_end=170),
]
sd = synth.make_state_dict(7)
for prec in ("fp32", "bf16"):
    for c in CASES:
        cfg = load_model_config(overrides={"time_step_end": c["step_end"], "n_joint_future": c["k"], "operand_precision": prec})
        eng = HipEngine(cfg)
        eng.load_state_dict(sd)
        batch = synth.make_batch(5000, c["b"], n_agent=c["a"], n_pl=c["p"], n_tl=40)
        s = scene_from_batch(batch, eng.device)
        enc = eng.encode_scene(s)
        feats = {"map_feature": enc["map_feature"], "map_feature_valid": enc["map_feature_valid"], "tl_feature": enc["tl_feature"]}
        n = c["b"] * c["k"]
        z = enc["latent_mean"].repeat_interleave(c["k"], 0).contiguous()
        if c["k"] > 1:
            z = z + 0.3 * torch.from_numpy(synth.make_latent_noise(1, n, c["a"])).to(z.device)
        dest = enc["dest_logits"].argmax(-1).to(torch.int32).repeat_interleave(c["k"], 0).contiguous()
        gv = s["agent_valid"].bool().any(1).to(torch.uint8).repeat_interleave(c["k"], 0).contiguous()
        out = None
        for _ in range(2):
            out = eng.rollout(s, feats, z, enc["latent_mean"], dest, gv, c["k"], c["step_end"], out=out)
        torch.cuda.synchronize()
        reps = 3
        t0 = time.perf_counter()
        for _ in range(reps):
            out = eng.rollout(s, feats, z, enc["latent_mean"], dest, gv, c["k"], c["step_end"], out=out)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        print(json.dumps({"case": c["name"], "operand_precision": prec, "instances": n, "ms_per_rollout": dt * 1e3,
                          "scene_steps_per_s": n * c["step_end"] / dt, "agent_steps_per_s": n * c["step_end"] * c["a"] / dt,
                          "finite": bool(torch.isfinite(out["preds"]).all())}))
        del eng
