"""Write synthetic WOMD-shaped episodes in the reference's packed-h5 format (`src/pack_h5_womd.py:378-392`) through
libtrafficbots_h5.so, for runs of `DataH5womd` / `validation_step` / `test_step` without the Waymo data:

    python tools/pack_synth_h5.py OUT_DIR [--episodes 64] [--agents 64] [--polylines 1024] [--stop-points 40] [--seed 500]

writes OUT_DIR/{training,validation,testing}.h5 (the same episodes: a training file holds the 91-step tensors, a testing file the
11-step history, a validation file both) and prints the `DataH5womd(...)` call that reads them."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trafficbots_amd import data_h5, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("out_dir")
ap.add_argument("--episodes", type=int, default=64)
ap.add_argument("--agents", type=int, default=64)
ap.add_argument("--polylines", type=int, default=1024)
ap.add_argument("--stop-points", type=int, default=40)
ap.add_argument("--seed", type=int, default=500)
a = ap.parse_args()
os.makedirs(a.out_dir, exist_ok=True)
episodes, attrs = synth.make_h5_episodes(a.seed, a.episodes, n_agent=a.agents, n_pl=a.polylines, n_tl=a.stop_points, p_invalid_agent=0.2,
                                         p_late_spawn=0.2, pos_range=140.0)
dm = data_h5.DataH5womd(a.out_dir, n_agent=a.agents, n_pl=a.polylines, n_tl_stop=a.stop_points)
data_h5.write_packed_h5(f"{a.out_dir}/validation.h5", episodes, attrs)
data_h5.write_packed_h5(f"{a.out_dir}/training.h5", [{k: e[k] for k in dm.tensor_size_train} for e in episodes])
data_h5.write_packed_h5(f"{a.out_dir}/testing.h5", [{k: e[k] for k in dm.tensor_size_test} for e in episodes], attrs)
n_no_sim, n_lane = episodes[0]["agent_no_sim/valid"].shape[1], episodes[0]["tl_lane/valid"].shape[1]
for f in ("training", "validation", "testing"):
    print(f"{a.out_dir}/{f}.h5: {os.path.getsize(f'{a.out_dir}/{f}.h5') / 2**20:.1f} MiB")
print(f"read with: DataH5womd({a.out_dir!r}, batch_size=..., n_agent={a.agents}, n_pl={a.polylines}, n_tl_stop={a.stop_points})\n"
      f"(the tensors the hot path does not read are smaller than Waymo's here: {n_no_sim} agent_no_sim, {n_lane} tl_lane; "
      f"`read_reference_batch` callers set those sizes in dm.tensor_size_* first)")
