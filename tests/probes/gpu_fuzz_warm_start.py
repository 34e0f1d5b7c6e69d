"""Development probe (GPU box): tb_rollout_io.warm_start_steps over randomly drawn shapes (agents around the 16-row tile and 32-key block
boundaries, empty traffic lights, K futures, late spawns): the batched warm start must equal the step-by-step rollout BIT FOR BIT in
every output, for several warm-start lengths.  No oracle involved (a property of the HIP path alone).
    python tests/probes/gpu_fuzz_warm_start.py [n_cases]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from trafficbots_amd import synth  # noqa: E402
from trafficbots_amd.config import load_model_config  # noqa: E402
from trafficbots_amd.runtime import HipEngine, scene_from_batch  # noqa: E402

torch.set_num_threads(min(8, torch.get_num_threads()))  # (the CPU oracles: small matrices, see tests/probes/oracle_thread_timing.py)
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.default_rng(int(os.environ.get("FUZZ_SEED", "909")))
dev = torch.device("cuda:0")
for ci in range(n_cases):
    a, p, t = int(rng.choice([1, 2, 15, 16, 17, 31, 33, 48, 64, 65, 80])), int(rng.choice([1, 16, 33, 64, 130, 256])), int(rng.choice([1, 8, 33, 40]))
    b, k, step_end = int(rng.integers(1, 7)), int(rng.integers(1, 4)), int(rng.choice([12, 20, 40]))
    scene = dict(n_agent=a, n_pl=p, n_tl=t, p_invalid_agent=float(rng.choice([0.0, 0.3, 0.8])), p_late_spawn=float(rng.choice([0.0, 0.4])),
                 p_invalid_pl=float(rng.choice([0.0, 0.4])), p_tl_valid=float(rng.choice([0.0, 0.5, 1.0])), pos_range=float(rng.choice([30.0, 140.0])))
    eng = HipEngine(load_model_config(overrides={"time_step_end": step_end, "n_joint_future": k}), "cuda:0")
    eng.load_state_dict(synth.make_state_dict(80000 + ci))
    s = scene_from_batch(synth.make_batch(81000 + ci, b, **scene), dev)
    assert s["warm_ok"]
    enc = eng.encode_scene(s)
    feats = {"map_feature": enc["map_feature"], "map_feature_valid": enc["map_feature_valid"], "tl_feature": enc["tl_feature"]}
    z = enc["latent_mean"].repeat_interleave(k, 0) + 0.3 * torch.from_numpy(synth.make_latent_noise(82000 + ci, b * k, a)).cuda()
    dest = enc["dest_logits"].argmax(-1).to(torch.int32).repeat_interleave(k, 0)
    gv = s["agent_valid"].bool().any(1).to(torch.uint8).repeat_interleave(k, 0)
    outs = []
    for w in (-1, 10, int(rng.integers(1, 10))):
        s2 = dict(s, warm_ok=False) if w < 0 else s
        o = eng.rollout(s2, feats, z, enc["latent_mean"], dest, gv, k, step_end, warm_start_steps=max(w, 0), tap_step=int(rng.integers(1, 12)))
        torch.cuda.synchronize()
        outs.append({key: v.clone() for key, v in o.items() if torch.is_tensor(v) and not key.startswith("tap")})
    bad = [key for other in outs[1:] for key, v in outs[0].items() if not torch.equal(v, other[key])]
    print(f"case {ci:2d} A={a:2d} P={p:3d} T={t:2d} B={b} K={k} S={step_end}  {'bit-identical' if not bad else 'DIFFERS: ' + ', '.join(sorted(set(bad)))}", flush=True)
    if bad:
        print(scene)
        sys.exit(1)
print(f"all {n_cases} cases bit-identical")
