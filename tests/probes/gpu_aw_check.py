"""Development probe (GPU box): the eight-wave assist carve of the bf16 step launch (TB_STEP_AW, tb::xba) against the four-wave
kernel on the stress shape (A = 128, P = 1024): same masks / flags, trajectories within bf16 rounding of each other at short
horizon, repeatable bit for bit, and the time of a rollout for both.  usage: python tests/probes/gpu_aw_check.py [step_end]"""
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

step_end = int(sys.argv[1]) if len(sys.argv) > 1 else 40
b, a, p = int(os.environ.get("AB_B", 32)), int(os.environ.get("AB_A", 128)), int(os.environ.get("AB_P", 1024))
kw = dict(n_agent=a, n_pl=p, n_tl=40)
if os.environ.get("AB_MASKS"):
    kw.update(p_invalid_agent=0.2, p_invalid_pl=0.4, p_late_spawn=0.2)
cfg = load_model_config(overrides={"time_step_end": step_end, "operand_precision": "bf16"})
sd = synth.make_state_dict(7)
batch = synth.make_batch(15000, b, **kw)
res = {}
for aw in ("1", "0", "1"):
    os.environ["TB_STEP_AW"] = aw
    eng = HipEngine(cfg)
    eng.load_state_dict(sd)
    s = scene_from_batch(batch, eng.device)
    enc = eng.encode_scene(s)
    feats = {"map_feature": enc["map_feature"], "map_feature_valid": enc["map_feature_valid"], "tl_feature": enc["tl_feature"]}
    z = enc["latent_mean"].clone()
    dest = enc["dest_logits"].argmax(-1).to(torch.int32)
    gv = s["agent_valid"].bool().any(1).to(torch.uint8)
    out = eng.rollout(s, feats, z, enc["latent_mean"], dest, gv, 1, step_end)
    torch.cuda.synchronize()
    eng.check_status()
    t0 = time.perf_counter()
    for _ in range(3):
        out = eng.rollout(s, feats, z, enc["latent_mean"], dest, gv, 1, step_end, out=out)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    key = aw if aw not in res else aw + "b"
    res[key] = {k: v.clone() for k, v in out.items() if torch.is_tensor(v)}
    print(json.dumps({"TB_STEP_AW": aw, "ms_per_rollout": dt * 1e3, "scene_steps_per_s": b * step_end / dt,
                      "finite": bool(torch.isfinite(out["preds"]).all())}), flush=True)
    del eng
r1, r0, r1b = res["1"], res["0"], res["1b"]
assert all(torch.equal(r1[k], r1b[k]) for k in r1), "the assist carve is not repeatable"
v = r0["valid"].bool()
same_valid = bool((r1["valid"] == r0["valid"]).all())
d = ((r1["preds"] - r0["preds"]).abs() * v.unsqueeze(-1))[..., :2]
per_step = d.amax(dim=(0, 1, 3))
print(json.dumps({"repeatable": True, "same_valid": same_valid, "max_xy_diff_step1": float(per_step[0]), "max_xy_diff_step11": float(per_step[min(11, len(per_step) - 1)]),
                  "max_xy_diff_last": float(per_step[-1]), "max_abs_pred": float(r0["preds"].abs().max())}))
