"""Development probe: run the same small rollout with the normal library and with the -DTB_DEBUG_LATE_TILE build (last row tile of every
instance delayed by ~200 us inside each launch) and compare: any difference means the tiles of an instance exchange data through a
buffer that a sibling running ahead may already have overwritten."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    sys.path.insert(0, ROOT)
    import torch
    from trafficbots_amd import synth
    from trafficbots_amd.waymo_motion import WaymoMotion

    wm = WaymoMotion(time_step_end=25, n_joint_future=2)
    wm.load_state_dict(synth.make_state_dict(11))
    batch = synth.make_batch(4300, 2, n_agent=48, n_pl=40, n_tl=8, p_late_spawn=0.2)
    eps = synth.make_latent_noise(4301, 4, 48)
    import time
    for _ in range(2):
        t0 = time.perf_counter()
        out = wm.test_step(batch, latent_eps=torch.from_numpy(eps).cuda(), generator=torch.Generator(device="cuda").manual_seed(3))
        torch.cuda.synchronize()
        print(f"[{os.path.basename(os.environ.get('TB_HIP_LIB', ''))}] test_step {1e3 * (time.perf_counter() - t0):.2f} ms", flush=True)
    np.savez(sys.argv[1], preds=out["rollout_buffer"].preds.cpu().numpy(), valid=out["rollout_buffer"].valid.cpu().numpy())
    sys.exit(0)
res = {}
for name, lib in (("normal", "libtrafficbots_hip.so"), ("late", "libtrafficbots_hip_late.so")):
    env = dict(os.environ, TB_HIP_LIB=os.path.join(ROOT, "trafficbots_amd", "lib", lib))
    subprocess.run([sys.executable, os.path.abspath(__file__), f"/tmp/late_{name}.npz"], check=True, env=env)
    res[name] = np.load(f"/tmp/late_{name}.npz")
d = np.abs(res["normal"]["preds"] - res["late"]["preds"]).max()
print(f"max |normal - late-tile| = {d:.3e}; valid equal: {bool((res['normal']['valid'] == res['late']['valid']).all())}")
sys.exit(0 if d == 0.0 else 1)
