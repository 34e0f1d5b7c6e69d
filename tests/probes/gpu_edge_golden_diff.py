import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from conftest import golden_inputs, load_golden
from trafficbots_amd.waymo_motion import WaymoMotion
g, meta = load_golden("edge_scenes")
cfg, sd, batch, eps = golden_inputs(meta)
wm = WaymoMotion(time_step_end=meta["time_step_end"], n_joint_future=meta["k"])
wm.load_state_dict(sd)
gs = torch.from_numpy(np.transpose(g["goal_sample"], (0, 2, 1)).copy())
out = wm.test_step(batch, latent_eps=torch.from_numpy(eps).cuda(), goal_sample=gs.cuda())
lgraw = out["dest_logits"].cpu().numpy()
lg = torch.log_softmax(out["dest_logits"], -1).cpu().numpy()
fin = np.isfinite(g["dest_logits"])
av = np.asarray(batch["history/agent/valid"]); mv = np.asarray(batch["map/valid"])
for b in range(lg.shape[0]):
    mism = (np.isfinite(lg[b]) != fin[b])
    print("scene", b, "agents valid(any)", int(av[b].any(0).sum()), "pl valid", int(mv[b].any(-1).sum()), "mismatch", int(mism.sum()),
          "| golden finite", int(fin[b].sum()), "hip raw finite", int(np.isfinite(lgraw[b]).sum()), "hip logsm finite", int(np.isfinite(lg[b]).sum()))
    if mism.any():
        a = np.argwhere(mism)[0]
        print("   first at agent", a[0], "pl", a[1], "golden", g["dest_logits"][b, a[0], :6], "hip raw", lgraw[b, a[0], :6], "agent valid hist", av[b, :, a[0]].astype(int))
buf = out["rollout_buffer"]
for k in ("valid", "preds", "action_log_probs"):
    gv = g[k]; hv = getattr(buf, k).cpu().numpy()
    print(k, gv.shape, hv.shape)
