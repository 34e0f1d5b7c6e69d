import sys, json, torch
sys.path.insert(0, '.')
import bench
from trafficbots_amd import synth
sd = synth.make_state_dict(7)
dev = torch.device('cuda', 0)
for name, spec in (("k2_full", dict(k=2, a=64, p=256, step_end=90, prec="fp32", what="256 tiles: full carve, no helpers")),
                   ("k3_lean", dict(k=3, a=64, p=256, step_end=90, prec="fp32", what="384 tiles: LEAN carve")),
                   ("k1_helpers", dict(k=1, a=64, p=256, step_end=90, prec="fp32", what="128 tiles: full carve + helpers"))):
    r = bench.sub_record(name, spec, sd, dev, 0, 1, 8, 3)
    print(name, {k: r[k] for k in ("value", "ms_per_pass", "k_step_fused_us", "host_cpu_ms_per_pass", "rollout_graph")})
