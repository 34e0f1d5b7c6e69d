import os, sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
from trafficbots_amd import synth
from trafficbots_amd.waymo_motion import WaymoMotion
torch.set_num_threads(min(8, torch.get_num_threads()))  # (the CPU oracles: small matrices, see tests/probes/oracle_thread_timing.py)
rng = np.random.default_rng(int(os.environ.get("FUZZ_SEED", "7")))
for ci in range(16):
    a, p, t = int(rng.choice([1, 2, 15, 16, 17, 33, 64, 65])), int(rng.choice([1, 2, 31, 32, 33, 96, 130])), int(rng.choice([1, 2, 31, 33, 40]))
    k, b, se = int(rng.integers(1, 4)), int(rng.integers(1, 4)), 40
    sd = synth.make_state_dict(300 + ci)
    batch = synth.make_batch(300 + ci, b, n_agent=a, n_pl=p, n_tl=t, p_invalid_agent=float(rng.choice([0.0, 0.5])), p_late_spawn=0.2)
    outs = {}
    for prec in ("fp32", "bf16"):
        wm = WaymoMotion(time_step_end=se, n_joint_future=k, operand_precision=prec, **{"traffic_rule_checker": {f"enable_check_{c}": True for c in ("collided", "run_road_edge", "run_red_light", "passive")}})
        wm.load_state_dict(sd)
        eps = torch.from_numpy(synth.make_latent_noise(1, b * k, a)).cuda()
        o = wm.test_step(batch, latent_eps=eps, generator=torch.Generator(device="cuda").manual_seed(1))
        torch.cuda.synchronize()
        outs[prec] = o["rollout_buffer"]
    f32, b16 = outs["fp32"], outs["bf16"]
    fin = bool(torch.isfinite(b16.preds).all() and torch.isfinite(f32.preds).all())
    d10 = float((f32.preds[..., :11, :2] - b16.preds[..., :11, :2]).abs().max())
    print(f"case {ci} A={a} P={p} T={t} K={k} B={b}: finite={fin} |fp32-bf16| first 11 steps {d10:.2e} collided {int(f32.violations['collided'][..., -1].sum())}/{int(b16.violations['collided'][..., -1].sum())}", flush=True)
    assert fin
print("ok")
