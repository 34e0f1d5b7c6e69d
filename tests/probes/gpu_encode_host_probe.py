import os, sys, time, torch
sys.path.insert(0, "/root/repo")
from trafficbots_amd import synth
from trafficbots_amd.waymo_motion import WaymoMotion
batch = synth.make_batch(5000, 32, n_agent=64, n_pl=256, n_tl=40)
wm = WaymoMotion(time_step_end=90, n_joint_future=1)
wm.load_state_dict(synth.make_state_dict(7))
for _ in range(3): wm.test_step(batch)
scene = wm.pre_processing(batch)
def seg(name, fn, n=10):
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): r=fn()
    t_host=(time.perf_counter()-t0)/n
    torch.cuda.synchronize(); t_all=(time.perf_counter()-t0)/n
    print(f"{name:36s} host {t_host*1e3:6.2f} ms  incl GPU {t_all*1e3:6.2f} ms")
seg("engine.encode_scene", lambda: wm.engine.encode_scene(scene))
seg("model.encode_input_features", lambda: wm.model.encode_input_features(scene))
e = wm.engine.encode_scene(scene)
seg("3x .bool()", lambda: (scene["agent_valid"].bool(), e["map_feature_valid"].bool(), scene["tl_valid"].bool()))
seg("encode_scene + sync each", lambda: (wm.engine.encode_scene(scene), torch.cuda.synchronize()), n=5)
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(10): wm.model.encode_input_features(scene)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(12)
