"""N > 1 path on CPU: two gloo ranks each roll out their shard of a seeded global batch (with the oracle
standing in for the GPU engine) and all-reduce the metric partials; the result must equal the single-process
run over the whole batch."""
import os
import socket

import numpy as np
import torch
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _partials_for(lo, hi):
    import sys

    sys.path.insert(0, ROOT)
    from oracle.trafficbots_oracle import Oracle
    from trafficbots_amd import synth
    from trafficbots_amd.config import load_model_config
    from trafficbots_amd.shard import metric_partials

    step_end = 14
    cfg = load_model_config(overrides={"time_step_end": step_end, "n_joint_future": 1})
    sd = synth.make_state_dict(3)
    batch = synth.make_batch(800, hi - lo, scene_offset=lo, n_agent=6, n_pl=12, n_tl=4, p_late_spawn=0.3, pos_range=148.0)
    with torch.no_grad():
        r = Oracle(sd, cfg, torch.float32, hoist=True).joint_future_pred(batch, 1, None, step_end)
    sq = lambda x: x[:, :, 0]  # noqa: E731  [B,A,K=1,S,..] -> [B,A,S,..]
    return metric_partials(sq(r["preds"]), sq(r["valid"]), sq(r["outside_map"]), sq(r["dest_reached"]), hi - lo, step_end)


def _worker(rank, world, port, n_global, q):
    import torch.distributed as dist

    torch.set_num_threads(1)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from trafficbots_amd.shard import all_reduce_partials, shard_range

    lo, hi = shard_range(n_global, rank, world)
    part = _partials_for(lo, hi)
    red, t = all_reduce_partials(part, elapsed_s=1.0 + rank)
    if rank == 0:
        q.put((red, t))
    dist.destroy_process_group()


def test_two_rank_gloo_matches_single_process():
    from trafficbots_amd.shard import PARTIAL_FIELDS, shard_range

    n_global, world = 5, 2
    assert shard_range(5, 0, 2) == (0, 3) and shard_range(5, 1, 2) == (3, 5)
    assert shard_range(256, 7, 8) == (224, 256)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_global, q)) for r in range(world)]
    for p in procs:
        p.start()
    red, t = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    whole = _partials_for(0, n_global)
    for i, k in enumerate(PARTIAL_FIELDS):
        assert abs(red[k] - float(whole[i])) <= 1e-6 * max(1.0, abs(float(whole[i]))), k
    assert t == 2.0  # MAX over ranks of the elapsed time
