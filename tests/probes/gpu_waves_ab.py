"""A/B probe (GPU box): the 4-wave and the 8-wave step kernels on the same inputs in one process -- result difference and time."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from trafficbots_amd import synth  # noqa: E402
from trafficbots_amd.config import load_model_config  # noqa: E402
from trafficbots_amd.runtime import HipEngine, scene_from_batch  # noqa: E402

shape = dict(b=int(os.environ.get("AB_B", 32)), a=int(os.environ.get("AB_A", 64)), p=int(os.environ.get("AB_P", 256)), s=int(os.environ.get("AB_S", 90)))
prec = os.environ.get("AB_PREC", "fp32")
sd = synth.make_state_dict(7)
cfg = load_model_config(overrides={"time_step_end": shape["s"], "n_joint_future": 1, "operand_precision": prec})
batch = synth.make_batch(5000, shape["b"], n_agent=shape["a"], n_pl=shape["p"], n_tl=40)
res = {}
for waves in ("4", "8"):
    os.environ["TB_STEP_WAVES"] = waves
    eng = HipEngine(cfg)
    eng.load_state_dict(sd)
    s = scene_from_batch(batch, eng.device)
    enc = eng.encode_scene(s)
    feats = {"map_feature": enc["map_feature"], "map_feature_valid": enc["map_feature_valid"], "tl_feature": enc["tl_feature"]}
    z = enc["latent_mean"].clone()
    dest = enc["dest_logits"].argmax(-1).to(torch.int32)
    gv = s["agent_valid"].bool().any(1).to(torch.uint8)
    out = None
    for _ in range(3):
        out = eng.rollout(s, feats, z, enc["latent_mean"], dest, gv, 1, shape["s"], out=out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 10
    for i in range(n):
        if i == n - 1:
            eng.set_timing(True)
        out = eng.rollout(s, feats, z, enc["latent_mean"], dest, gv, 1, shape["s"], out=out)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    tm = eng.get_timing()
    res[waves] = dict(preds=out["preds"].clone(), valid=out["valid"].clone(), ms=dt * 1e3, k_us=tm["fused_ms"] / max(1, tm["n_fused"]) * 1e3,
                      edge_ms=tm["edge_ms"])
a, b = res["4"], res["8"]
d = (a["preds"] - b["preds"]).abs() * (a["valid"] & b["valid"]).unsqueeze(-1)
print(json.dumps({"shape": shape, "prec": prec, "ms_4": a["ms"], "ms_8": b["ms"], "k_us_4": a["k_us"], "k_us_8": b["k_us"], "edge_4": a["edge_ms"],
                  "edge_8": b["edge_ms"], "bitwise_equal": bool(torch.equal(a["preds"], b["preds"])), "valid_equal": bool(torch.equal(a["valid"], b["valid"])),
                  "max_abs_xy_diff": float(d[..., :2].max()), "max_abs_xy_diff_step20": float(d[:, :, :20, :2].max())}))
