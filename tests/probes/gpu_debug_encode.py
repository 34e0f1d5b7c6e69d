"""Development probe: encoder outputs of tb_encode_scene vs the oracle."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.trafficbots_oracle import Oracle
from trafficbots_amd import synth
from trafficbots_amd.config import load_model_config
from trafficbots_amd.runtime import HipEngine, scene_from_batch
cfg = load_model_config()
sd = synth.make_state_dict(7)
for (b, a, p) in ((1, 8, 32), (2, 64, 256)):
    batch = synth.make_batch(1000, b, n_agent=a, n_pl=p, n_tl=40)
    orc = Oracle(sd, cfg, torch.float32)
    with torch.no_grad():
        inp = orc.preprocess(batch); f = orc.encode_scene(inp); mean, _, lv = orc.latent_prior(f); lg = orc.dest_logits(f, inp)
    eng = HipEngine(cfg); eng.load_state_dict(sd)
    s = scene_from_batch(batch, torch.device("cuda:0"))
    e = eng.encode_scene(s); torch.cuda.synchronize()
    print(f"B={b} A={a} P={p}")
    print("  agent_feature", (e["agent_feature"].cpu() - f["agent_feature"]).abs().max().item())
    print("  tl_feature   ", (e["tl_feature"].cpu() - f["tl_feature"]).abs().max().item())
    print("  map_feature  ", (e["map_feature"].cpu() - f["map_feature"]).abs().max().item(), "nan:", torch.isnan(e["map_feature"]).sum().item())
    print("  latent_mean  ", (e["latent_mean"].cpu() - mean).abs().max().item(), "nan:", torch.isnan(e["latent_mean"]).sum().item())
    fin = torch.isfinite(lg)
    d = torch.where(fin, e["dest_logits"].cpu() - lg, torch.zeros_like(lg)).abs()
    print("  dest_logits  ", d.max().item(), "finite-mask equal:", bool((torch.isfinite(e["dest_logits"].cpu()) == fin).all()))
