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


def _worker8(rank, world, port, n_global, q):
    """one of 8 ranks: uneven shards (a rank may own NO scene), per-rank tails of the one all-reduce, collective count"""
    import torch.distributed as dist

    torch.set_num_threads(1)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from trafficbots_amd import shard

    lo, hi = shard.shard_range(n_global, rank, world)
    if hi > lo:
        part = _partials_for(lo, hi)
    else:  # an idle rank still takes part in the collective, with zero partials
        part = torch.zeros(len(shard.PARTIAL_FIELDS), dtype=torch.float64)
    n0 = shard.N_COLLECTIVES
    red, t, ranks = shard.all_reduce_partials(part, elapsed_s=1.0 + 0.25 * rank, per_rank={"device_plus_1": rank + 1, "n_scene": hi - lo})
    q.put((rank, red, t, ranks, shard.N_COLLECTIVES - n0))
    dist.destroy_process_group()


def test_eight_rank_gloo_uneven_shards_and_idle_rank():
    """The 8-rank flow of BASELINE configs[2] on CPU (gloo): shard_range over 8 ranks for an uneven global batch (250 scenes: 31 or 32
    per rank) and for fewer scenes than ranks (6 over 8: two idle ranks), ONE collective, every rank sees every rank's slot."""
    from trafficbots_amd.shard import PARTIAL_FIELDS, shard_range

    world = 8
    blocks = [shard_range(250, r, world) for r in range(world)]
    assert blocks[0] == (0, 32) and blocks[1] == (32, 64) and blocks[2] == (64, 95) and blocks[-1] == (219, 250)
    assert all(b[1] == c[0] for b, c in zip(blocks, blocks[1:])) and sorted({hi - lo for lo, hi in blocks}) == [31, 32]
    n_global = 6
    assert [shard_range(n_global, r, world) for r in (5, 6, 7)] == [(5, 6), (6, 6), (6, 6)]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker8, args=(r, world, port, n_global, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    whole = _partials_for(0, n_global)
    for rank, red, t, ranks, n_coll in got:
        assert n_coll == 1, (rank, n_coll)
        for i, k in enumerate(PARTIAL_FIELDS):
            assert abs(red[k] - float(whole[i])) <= 1e-6 * max(1.0, abs(float(whole[i]))), (rank, k)
        assert t == 1.0 + 0.25 * 7
        assert ranks["elapsed_s"] == [1.0 + 0.25 * r for r in range(world)]
        assert ranks["device_plus_1"] == [float(r + 1) for r in range(world)]
        assert ranks["n_scene"] == [1.0] * 6 + [0.0, 0.0]


def _val_states_for(lo, hi, batch=None):
    """TrainingMetrics sum-states of scenes [lo, hi) of a seeded validation batch, or of `batch` (oracles standing in for the GPU engine)."""
    import sys

    sys.path.insert(0, ROOT)
    from oracle import training_oracle as TO
    from oracle.trafficbots_oracle import Oracle
    from trafficbots_amd import synth
    from trafficbots_amd.config import load_model_config
    from trafficbots_amd.runtime import TRAIN_FIELDS

    step_end = 20
    cfg = load_model_config(overrides={"time_step_end": step_end, "n_joint_future": 1})
    sd = synth.make_state_dict(4)
    if batch is None:
        batch = synth.make_val_batch(900, hi - lo, scene_offset=lo, n_agent=6, n_pl=12, n_tl=4, p_future_exit=0.3)
    with torch.no_grad():
        r = Oracle(sd, cfg, torch.float32).reactive_replay(batch, step_end)
    gv = r["gt_valid"][:, 1: step_end + 1].transpose(1, 2)
    gs = r["gt_state"][:, 1: step_end + 1].transpose(1, 2)
    rew, rv = TO.differentiable_reward(r["valid"], r["preds"], gv, gs, r["agent_size"], cfg["differentiable_reward"])
    st = TO.training_metric_states(r["valid"], rv, rew, r["override_masks"], r["agent_role"], r["dest_logits_raw"], r["goal_valid"],
                                   r["gt_dest"], r["post_mean"], r["post_log_std"], r["post_valid"], r["prior_mean"],
                                   r["prior_log_std"], r["prior_valid"], cfg["training_metrics"])
    return torch.tensor([st[k] for k in TRAIN_FIELDS], dtype=torch.float64), cfg


def _val_worker(rank, world, port, n_global, q):
    import torch.distributed as dist

    torch.set_num_threads(1)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from trafficbots_amd.metrics import TrainingMetrics
    from trafficbots_amd.shard import shard_range

    lo, hi = shard_range(n_global, rank, world)
    st, cfg = _val_states_for(lo, hi)
    m = TrainingMetrics("reactive_replay", **cfg["training_metrics"])
    m.update(st)
    m.sync()  # the validation path's collective: one SUM all-reduce of the packed states
    if rank == 0:
        q.put(m.compute())
    dist.destroy_process_group()


def test_two_rank_gloo_validation_losses_match_single_process():
    from trafficbots_amd.metrics import TrainingMetrics

    n_global, world = 3, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_val_worker, args=(r, world, port, n_global, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    st, cfg = _val_states_for(0, n_global)
    m = TrainingMetrics("reactive_replay", **cfg["training_metrics"])
    m.update(st)
    want = m.compute()
    assert set(got) == set(want) == {"reactive_replay/loss", "reactive_replay/vae_kl", "reactive_replay/diffbar_reward",
                                     "reactive_replay/goal_loss"}
    for k in want:
        assert abs(got[k] - want[k]) <= 1e-5 * max(1.0, abs(want[k])), k


H5_SCENE = dict(n_agent=6, n_pl=12, n_tl=4, p_future_exit=0.3)


def _h5_module(data_dir, rank, world):
    from trafficbots_amd import data_h5, synth

    dm = data_h5.DataH5womd(data_dir, batch_size=2, n_agent=6, n_pl=12, n_tl_stop=4, rank=rank, world_size=world)
    ep = synth.make_h5_episodes(900, 1, **H5_SCENE)[0][0]
    for k in list(dm.tensor_size_val):
        dm.tensor_size_val[k] = ep[k].shape
    dm.setup("validate")
    return dm


def _h5_states(dm):
    """sum-states over the batches of this rank's loader (the oracle wants the reference's layout: re-read each batch that way)"""
    from trafficbots_amd import data_h5

    f = data_h5.PackedH5File(dm.path_val_h5)
    total, cfg = None, None
    for packed in dm.val_dataloader():
        ref = {k: (v.numpy() if torch.is_tensor(v) else v)
               for k, v in f.read_reference_batch(packed["episode_idx"].tolist(), dm.tensor_size_val, False).items()}
        st, cfg = _val_states_for(0, 0, batch=ref)
        total = st if total is None else total + st
    return total, cfg


def _h5_worker(rank, world, port, data_dir, q):
    import sys

    import torch.distributed as dist

    sys.path.insert(0, ROOT)
    torch.set_num_threads(1)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from trafficbots_amd.metrics import TrainingMetrics

    st, cfg = _h5_states(_h5_module(data_dir, rank, world))
    m = TrainingMetrics("reactive_replay", **cfg["training_metrics"])
    m.update(st)
    m.sync()
    if rank == 0:
        q.put(m.compute())
    dist.destroy_process_group()


def test_two_rank_gloo_validation_from_a_packed_h5_file(tmp_path):
    """both ranks read their round-robin share of one packed file concurrently; the synced losses equal one process over the whole file"""
    from conftest import require_h5

    require_h5()
    from trafficbots_amd import data_h5, synth
    from trafficbots_amd.metrics import TrainingMetrics

    episodes, attrs = synth.make_h5_episodes(900, 5, **H5_SCENE)
    data_h5.write_packed_h5(str(tmp_path / "validation.h5"), episodes, attrs)
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_h5_worker, args=(r, world, port, str(tmp_path), q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    st, cfg = _h5_states(_h5_module(str(tmp_path), 0, 1))
    m = TrainingMetrics("reactive_replay", **cfg["training_metrics"])
    m.update(st)
    want = m.compute()
    st_mem, _ = _val_states_for(0, 5)  # and the file holds what synth.make_val_batch(900, 5) holds
    assert torch.allclose(st, st_mem, rtol=1e-6)
    for k in want:
        assert abs(got[k] - want[k]) <= 1e-5 * max(1.0, abs(want[k])), k


def _sync_worker(rank, world, port, q):
    import sys
    import torch.distributed as dist

    sys.path.insert(0, ROOT)
    torch.set_num_threads(1)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from trafficbots_amd.metrics import ErrorMetrics

    m = ErrorMetrics("x")
    if rank == 0:  # rank 1 received no batch (PackedSceneLoader does not pad ranks): it must still take part in the collective
        m.update(torch.tensor([2.0, 4.0, 6.0, 8.0], dtype=torch.float64))
    m.sync()
    first = m.compute()
    m.sync()  # a second sync must not multiply anything by the world size
    second = m.compute()
    own = None if m.states is None else m.states.clone()
    if rank == 0:
        m.update(torch.tensor([2.0, 0.0, 0.0, 0.0], dtype=torch.float64))  # update after sync: the accumulator was never all-reduced in place
    m.sync()
    third = m.compute()
    q.put((rank, first, second, third, None if own is None else own.tolist()))
    dist.destroy_process_group()


def test_metric_sync_is_idempotent_and_tolerates_an_idle_rank():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sync_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict((r[0], r[1:]) for r in (q.get(timeout=120), q.get(timeout=120)))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank in (0, 1):
        first, second, third, own = got[rank]
        assert first == second == {"x/err/pos_meter": 2.0, "x/err/rot_deg": 3.0, "x/err/spd_m_per_s": 4.0}
        assert third == {"x/err/pos_meter": 1.0, "x/err/rot_deg": 1.5, "x/err/spd_m_per_s": 2.0}
    assert got[0][3] == [2.0, 4.0, 6.0, 8.0] and got[1][3] is None
