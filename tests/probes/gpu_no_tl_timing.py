"""Development probe (GPU box): rollout time of the headline shape with lit traffic lights (p_tl_valid=0.3) and without any
(p_tl_valid=0: as2tl keeps only its FFN halves, ffn_layer_x)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from trafficbots_amd import synth
from trafficbots_amd.config import load_model_config
from trafficbots_amd.runtime import HipEngine, scene_from_batch
cfg = load_model_config(overrides={"time_step_end": 90, "n_joint_future": 1})
eng = HipEngine(cfg, "cuda:0"); eng.load_state_dict(synth.make_state_dict(7))
for ptl in (0.3, 0.0):
    scene = scene_from_batch(synth.make_batch(5000, 32, n_agent=64, n_pl=256, n_tl=40, p_tl_valid=ptl), torch.device("cuda:0"))
    enc = eng.encode_scene(scene)
    feats = {"map_feature": enc["map_feature"], "map_feature_valid": enc["map_feature_valid"], "tl_feature": enc["tl_feature"]}
    z = enc["latent_mean"].clone(); dest = enc["dest_logits"].argmax(-1).to(torch.int32); gv = scene["agent_valid"].bool().any(1).to(torch.uint8)
    out = None
    for _ in range(5): out = eng.rollout(scene, feats, z, enc["latent_mean"], dest, gv, 1, 90, out=out)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): out = eng.rollout(scene, feats, z, enc["latent_mean"], dest, gv, 1, 90, out=out)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 30
    print(f"p_tl_valid={ptl}: {dt*1e3:.3f} ms per rollout, {32*90/dt:.0f} scene-steps/s")
