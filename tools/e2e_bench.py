"""End-to-end measurement of the DROP-IN path (VERDICT r05 task 1): `WaymoMotion.test_step(batch)` -- what the reference's harness
calls per batch (`src/pl_modules/waymo_motion.py:902-949`) -- with a FRESH host batch per call (distinct numpy batches in rotation:
nothing is resident, every call pays its staging, upload, encoders, samplers, rollout, rule checks and post-processing), at

  (i)   the headline shape, K = 1        (ii)  the headline shape, K = 6
  (iii) the real WOMD shape A = 64, P = 1024 fed by `PackedSceneLoader` from a packed-h5 file (`src/data_modules/data_h5_womd.py:85-171`)

each as plain `test_step(batch)` calls and through `wm.prefetch(loader)` (next batch staged + encoded under the current rollout).
Used by bench.py (the `e2e` block of the line, outside `value`) and by tests/probes/gpu_e2e_fresh.py.  Synthetic data, random-init
weights; GPU box only.
"""
from __future__ import annotations

import os
import sys
import tempfile
import time
from typing import Dict, List

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trafficbots_amd import synth  # noqa: E402


def _loop(wm, batches: List, n: int, prefetch: bool) -> float:
    """ms per batch over n calls (after 2 untimed ones), one synchronize at the end."""
    stream = [batches[i % len(batches)] for i in range(n + 2)]
    torch.cuda.synchronize()
    time.sleep(0.05)  # (steady state of THIS mode: contexts of the measurement before have gone quiet -- the library's helper
    #                    workgroups and L2 warmers stand down while a neighbour context launched within the last 25 ms)
    it = iter(wm.prefetch(stream)) if prefetch else iter(stream)
    wm.test_step(next(it))
    wm.test_step(next(it))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for b in it:
        wm.test_step(b)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def _loop_lanes(wm, batches: List, n: int, lanes: int) -> float:
    """ms per batch through `wm.pipeline(stream, lanes)` (results consumed in order, each range-checked before it is handed out)."""
    stream = [batches[i % len(batches)] for i in range(n + 2 * lanes)]
    it = iter(wm.pipeline(stream, lanes=lanes))
    for _ in range(2 * lanes):
        next(it)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in it:
        pass
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def _stages(wm, batches: List, n: int = 10) -> Dict[str, Dict[str, float]]:
    """Per-stage time of `test_step`, each stage run n times on rotating batches: `host_ms` = until the call returns (enqueue cost),
    `ms` = until the GPU has finished it (stage run alone, back to back)."""
    out = {}
    torch.cuda.synchronize()
    time.sleep(0.05)  # (the lanes measured before have gone quiet: see _loop)

    def seg(name, fn):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            r = fn(i)
        t_host = (time.perf_counter() - t0) / n * 1e3
        torch.cuda.synchronize()
        out[name] = {"host_ms": t_host, "ms": (time.perf_counter() - t0) / n * 1e3}
        return r

    st = wm.engine.stager(wm._tf_params)
    plan, info = st.plan(batches[0])
    slot, slab = st._slab(plan.nbytes)
    seg("host_pack_into_pinned_slab", lambda i: st.fill(plan.host_views(slab.numpy()[: plan.nbytes]), batches[i % len(batches)], info))
    scene = seg("pre_processing (pack + ONE upload)", lambda i: wm.pre_processing(batches[i % len(batches)]))
    scene.pop("gt", None)
    wm.model._scene = None
    feats = seg("encode_input_features (tb_encode_scene)", lambda i: (setattr(wm.model, "_scene", None), wm.model.encode_input_features(scene))[1])
    gv = wm.model._goal_valid()

    def jfp(i):
        return wm.joint_future_pred(scene, feats, wm.model.latent_encoder(), wm.model.goal_manager.pred_goal(), gv)

    buf, gs, glp = seg("joint_future_pred (samplers + tb_rollout + rule checks)", jfp)
    scores = torch.exp(buf.latent_log_probs[..., 0] + glp)
    seg("waymo_post_processing (tb_post_process)", lambda i: wm.waymo_post_processing(
        valid=buf.valid[:, :, 0].any(-1), scores=scores, trajs=buf.preds[:, :, :, buf.step_future_start:], agent_type=scene["agent_type"]))
    out["upload_bytes_per_batch"] = plan.nbytes
    return out


def _kernel_census(wm, batches: List) -> Dict:
    """Device kernels of ONE test_step by name prefix (torch.profiler over roctracer): the library's (`tb::`) against everything else
    (torch glue, copies).  None when the profiler is not available."""
    try:
        from torch.profiler import ProfilerActivity, profile

        wm.test_step(batches[0])
        torch.cuda.synchronize()
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            wm.test_step(batches[1 % len(batches)])
            torch.cuda.synchronize()
        names = [e.name for e in prof.events() if getattr(e, "device_type", None) is not None and str(e.device_type).endswith("CUDA")]
        if not names:
            return {"available": False}
        tb = [n for n in names if "tb::" in n]
        cp = [n for n in names if "copy" in n.lower() or "memcpy" in n.lower() or "fill" in n.lower() or "memset" in n.lower()]
        other = [n for n in names if n not in tb and n not in cp]
        return {"available": True, "library_kernels": len(tb), "copies_and_fills": len(cp), "other_kernels": len(other),
                "other_kernel_names": sorted(set(o[:70] for o in other))[:12]}
    except Exception as e:  # the census is a courtesy; profiles/r06_rocprof_e2e.txt holds the rocprofv3 trace
        return {"available": False, "error": f"{type(e).__name__}: {e}"[:200]}


def _case(sd, dev, k: int, batches: List, n: int, scenes: int, step_end: int = 90, stages: bool = False) -> Dict:
    from trafficbots_amd.waymo_motion import WaymoMotion

    wm = WaymoMotion(time_step_end=step_end, n_joint_future=k, device=str(dev))
    wm.load_state_dict(sd)
    rec: Dict = {"k_futures": k, "scenes_per_batch": scenes, "sim_steps": step_end, "distinct_host_batches": len(batches)}
    for name, pre, chk in (("plain", False, True), ("plain_no_range_check", False, False), ("prefetch", True, True),
                           ("prefetch_no_range_check", True, False)):
        wm.check_range = chk
        ms = _loop(wm, batches, n, pre)
        rec[name] = {"ms_per_batch": ms, "scene_steps_per_s": scenes * step_end * k / (ms * 1e-3)}
    wm.check_range = True
    for lanes in (2, 3):
        ms = _loop_lanes(wm, batches, 2 * n, lanes)
        rec[f"pipeline_{lanes}_lanes"] = {"ms_per_batch": ms, "scene_steps_per_s": scenes * step_end * k / (ms * 1e-3)}
    rec["range_check_cost_ms"] = rec["plain"]["ms_per_batch"] - rec["plain_no_range_check"]["ms_per_batch"]
    if stages:
        wm.check_range = False
        rec["stages"] = _stages(wm, batches)
        rec["kernel_census_one_step"] = _kernel_census(wm, batches)
    st = wm.engine.stager(wm._tf_params)
    rec["uploads_per_batch"] = 1
    rec["stager"] = {"uploads": st.n_uploads, "bytes": st.bytes_uploaded}
    del wm
    return rec


def _womd_file(path: str, n_episode: int, n_pl: int) -> Dict:
    from trafficbots_amd import data_h5

    scene = dict(n_agent=64, n_pl=n_pl, n_tl=40)
    episodes, attrs = synth.make_h5_episodes(6100, n_episode, **scene)
    test_eps = [{k: v for k, v in e.items() if k.startswith(("history/", "map/"))} for e in episodes]
    data_h5.write_packed_h5(path, test_eps, attrs, deflate=0)
    return {k: v.shape for k, v in test_eps[0].items()}


def measure_rank(sd, dev, rank: int, n: int = 6) -> Dict:
    """The N > 1 form (every rank runs it at the same time: eight launching threads, eight stagers on one host): plain calls and the
    two-lane pipeline at the headline shape, this rank's own batches."""
    from trafficbots_amd.waymo_motion import WaymoMotion

    batches = [synth.make_batch(5000 + 37 * i + 1009 * rank, 32, n_agent=64, n_pl=256, n_tl=40) for i in range(4)]
    wm = WaymoMotion(time_step_end=90, n_joint_future=1, device=str(dev))
    wm.load_state_dict(sd)
    plain = _loop(wm, batches, n, False)
    lanes = _loop_lanes(wm, batches, 2 * n, 2)
    return {"plain_ms_per_batch": plain, "pipeline_2_lanes_ms_per_batch": lanes}


def measure(sd, dev, n: int = 12, womd: bool = True) -> Dict:
    """The `e2e` block of the bench line."""
    t_all = time.perf_counter()
    out: Dict = {"what": "WaymoMotion.test_step(batch) end to end, fresh host batch per call (4 distinct numpy batches in rotation); "
                         "plain = one call after the other, prefetch = `for sb in wm.prefetch(loader): wm.test_step(sb)` (next batch "
                         "staged + encoded on a side stream under the current rollout), pipeline_N_lanes = `for out in "
                         "wm.pipeline(loader, lanes=N)` (consecutive batches on N contexts / streams, results in order and "
                         "range-checked); ms per 32-scene batch",
                 "unit": "ms per batch; scene-steps/s"}
    batches = [synth.make_batch(5000 + 37 * i, 32, n_agent=64, n_pl=256, n_tl=40) for i in range(4)]
    out["headline_k1"] = _case(sd, dev, 1, batches, n, 32, stages=True)
    out["headline_k6"] = _case(sd, dev, 6, batches, max(6, n // 2), 32)
    if womd:
        try:
            from trafficbots_amd import data_h5
            from trafficbots_amd.waymo_motion import WaymoMotion

            with tempfile.TemporaryDirectory() as d:
                shapes = _womd_file(os.path.join(d, "testing.h5"), 64, 1024)
                dm = data_h5.DataH5womd(d, batch_size=32, n_agent=64, n_pl=1024, n_tl_stop=40)
                for key in list(dm.tensor_size_test):
                    dm.tensor_size_test[key] = shapes[key]
                dm.setup("test")
                wm = WaymoMotion(time_step_end=90, n_joint_future=1, device=str(dev))
                wm.load_state_dict(sd)
                rec = {"shape": "A = 64, P = 1024, T = 40 (the real WOMD tensor sizes), 64 episodes in a packed-h5 file, batches of 32 from PackedSceneLoader "
                                "(2 reader threads decode into one pinned slab per batch)", "k_futures": 1, "scenes_per_batch": 32}

                class Epochs:  # the loader over several epochs as one stream
                    def __init__(self, loader, n_epoch):
                        self.loader, self.n_epoch = loader, n_epoch

                    def __iter__(self):
                        for _ in range(self.n_epoch):
                            yield from self.loader

                    def __len__(self):
                        return self.n_epoch * len(self.loader)

                loader = dm.test_dataloader()
                n_ep = max(3, n // 2)
                for name, pre in (("plain", False), ("prefetch", True)):
                    it = iter(wm.prefetch(Epochs(loader, n_ep + 1)) if pre else Epochs(loader, n_ep + 1))
                    wm.test_step(next(it))
                    wm.test_step(next(it))
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    nb = 0
                    for b in it:
                        wm.test_step(b)
                        nb += 1
                    torch.cuda.synchronize()
                    ms = (time.perf_counter() - t0) / nb * 1e3
                    rec[name] = {"ms_per_batch": ms, "scene_steps_per_s": 32 * 90 / (ms * 1e-3), "batches": nb}
                it = iter(wm.pipeline(Epochs(loader, n_ep + 2), lanes=2))
                for _ in range(4):
                    next(it)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                nb = sum(1 for _ in it)
                torch.cuda.synchronize()
                ms = (time.perf_counter() - t0) / nb * 1e3
                rec["pipeline_2_lanes"] = {"ms_per_batch": ms, "scene_steps_per_s": 32 * 90 / (ms * 1e-3), "batches": nb}
                t0 = time.perf_counter()
                nb = sum(1 for _ in Epochs(loader, 2))
                rec["loader_alone_ms_per_batch"] = (time.perf_counter() - t0) / nb * 1e3
                out["womd_shape_packed_h5"] = rec
                del wm
        except Exception as e:
            out["womd_shape_packed_h5"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    out["seconds"] = time.perf_counter() - t_all
    return out


if __name__ == "__main__":
    import json

    print(json.dumps(measure(synth.make_state_dict(7), torch.device("cuda:0")), indent=1))
