"""Headline benchmark: closed-loop rollout throughput (scene-steps/s) on BASELINE.json configs[1]
(B=32 scenes x 64 agents x 256 polylines, 90 executed steps = 10 teacher-forced + 80 free, fp32),
scene-parallel over N GPUs (one process per GPU, one RCCL all-reduce of metric partials per pass).

One "step" of this bench = one pass of the hot path (tb_rollout: K/V hoists + the 90-step closed loop) over
one batch of 32 synthetic scenes per GPU whose encoded features are already resident in HBM; the one-time
scene encoders are timed separately (`encode_ms`).  Prints ONE JSON line (rank 0).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from trafficbots_amd import synth  # noqa: E402
from trafficbots_amd.config import load_model_config  # noqa: E402

B_PER_GPU, N_AGENT, N_PL, N_TL, STEP_END = 32, 64, 256, 40, 90
H = 128
PEAK_FP32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 dense peak
# fabric-side bytes per fused launch from the PMC pass of this round (profiles/r01_rocprof_xdl.txt: FETCH_SIZE 41 061 KiB x 2, the
# gfx950 correction of the guide, at the headline shape); collected with rocprofv3 --pmc in its own run, not at bench time
TRAFFIC_BYTES_PER_LAUNCH_B32 = 2 * 38870 * 1024


def flops_step_a(a, p, t):  # SURVEY 8(d) split: as2pl + as2tl + interaction K/V projections + agent encoder
    return 60 * a * H * H + 12 * a * H * (p + t) + 2 * a * (11 * 32 + 32 * 32)


def load_path(k_us, n_tl_keys=None):
    """Bytes every workgroup pulls through its CU's vector-memory path per fused launch (weights are streamed once per
    16-agent tile, K/V once per head) against the ~64 B/clk/CU the L1 can fill (MI355X_MICROARCH.md: L2 34.5 TB/s / 256 CUs)."""
    h = 128
    w_bytes = 65 * h * h * 4          # 67 H^2 weights of the path, 4 B each as an fp16 pair, minus the two constant half-Linears of
                                      # add_goal / add_latent that the rollout prologue hoists (k_fuse_hoist_x)
    pad = lambda n: (n + 31) // 32 * 32
    # (the hoist compacts the valid targets: the step walks pad32(valid) keys; all polylines / agents of the synthetic scenes are valid)
    kv_bytes = 3 * 2 * (pad(N_PL) + (pad(N_TL) if n_tl_keys is None else n_tl_keys) + pad(N_AGENT)) * h * 4
    clk = 2.1e9                       # s_memtime ticks per second observed on this kernel
    per_clk = (w_bytes + kv_bytes) / (k_us * 1e-6 * clk)
    return {"bytes_per_workgroup_launch": w_bytes + kv_bytes, "weights": w_bytes, "kv": kv_bytes, "achieved_B_per_clk_per_CU": per_clk,
            "peak_B_per_clk_per_CU": 64.0, "frac": per_clk / 64.0}


def flops_step_c(a):  # interaction attention/FFN + GRU + add_goal + add_latent + one action-head branch
    return 74 * a * H * H + 12 * a * H * a + 4 * a * H


def usable_cpus() -> int:
    """Cores this process may really use: affinity mask capped by the cgroup CPU quota (os.cpu_count() reports the
    host's cores and oversubscribing OpenMP threads on a quota-limited container is catastrophically slow)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, q // per))
        except Exception:
            pass
    return max(1, min(n, 64))


def cpu_baseline_worker(budget_s: float = 15.0) -> None:
    """Runs in a subprocess (hard timeout in the parent): the oracle (CPU port of the reference path, un-hoisted =
    the reference's op sequence) on a bounded sample sized from a short probe so that it takes ~budget_s."""
    from oracle.trafficbots_oracle import Oracle

    n_threads = usable_cpus()
    torch.set_num_threads(n_threads)
    cfg = load_model_config(overrides={"time_step_end": STEP_END, "n_joint_future": 1})
    sd = synth.make_state_dict(7)
    orc = Oracle(sd, cfg, torch.float32, hoist=False)

    def prep(b):
        batch = synth.make_batch(5000, b, n_agent=N_AGENT, n_pl=N_PL, n_tl=N_TL)
        inp = orc.preprocess(batch)
        f = orc.encode_scene(inp)
        mean, _, _ = orc.latent_prior(f)
        dest = orc.dest_logits(f, inp).argmax(-1)
        return inp, f, mean, dest, inp["agent_valid"].any(1)

    with torch.no_grad():
        inp, f, mean, dest, gv = prep(2)
        orc.rollout(inp, f, mean, mean, dest, gv, 1, 4)  # warm-up
        t0 = time.time()
        orc.rollout(inp, f, mean, mean, dest, gv, 1, 9)
        per_scene_step = (time.time() - t0) / (2 * 9)
        # batch efficiency improves with B; size the sample for ~budget_s at the probed rate (bounded 2..32 scenes)
        # (the small-batch probe under-estimates the rate ~3x: take the real workload's 32 scenes whenever affordable)
        b = B_PER_GPU if per_scene_step * STEP_END * B_PER_GPU <= 4 * budget_s else int(max(2, budget_s / (per_scene_step * STEP_END)))
        if b != 2:
            inp, f, mean, dest, gv = prep(b)
        # repeat the rollout until ~budget_s of CPU work has been timed (at least once, at most 8 times)
        reps, dt_ = 0, 0.0
        while reps < 1 or (dt_ < budget_s and reps < 8):
            t0 = time.time()
            orc.rollout(inp, f, mean, mean, dest, gv, 1, STEP_END)
            dt_ += time.time() - t0
            reps += 1
    print(json.dumps({
        "value": reps * b * STEP_END / dt_, "unit": "scene-steps/s", "cores": int(n_threads), "kind": "port",
        "sample": f"{reps} x oracle rollout (the reference's op sequence, PyTorch-CPU fp32, {n_threads} threads) of {b} scenes x {N_AGENT} "
                  f"agents x {N_PL} polylines x {STEP_END} steps in {dt_:.2f} s",
    }))


def cpu_baseline(timeout_s: float = 150.0):
    import subprocess

    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker"], capture_output=True, text=True,
                           timeout=timeout_s)
        for ln in reversed(r.stdout.strip().splitlines()):
            if ln.startswith("{"):
                return json.loads(ln)
        return {"value": None, "unit": "scene-steps/s", "cores": usable_cpus(), "kind": "port",
                "sample": f"worker failed: {r.stderr[-300:]}"}
    except subprocess.TimeoutExpired:
        return {"value": None, "unit": "scene-steps/s", "cores": usable_cpus(), "kind": "port",
                "sample": f"worker exceeded {timeout_s:.0f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--operand-precision", choices=["fp32", "bf16"], default="fp32",
                    help="bf16: BASELINE.json configs 4/5 operand precision (not the headline metric, which is fp32)")
    ap.add_argument("--cpu-baseline-worker", action="store_true")
    args = ap.parse_args()
    if args.cpu_baseline_worker:
        cpu_baseline_worker()
        return

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    # TB_BENCH_BACKEND=gloo is a dry-run hook: several ranks sharing the visible GPU(s) exercise the N > 1 control flow (shards,
    # barriers, max-over-ranks time, the all-reduce through host memory) where RCCL would refuse two ranks on one device
    backend = os.environ.get("TB_BENCH_BACKEND", "nccl")
    if backend == "gloo":
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "gloo":
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=dev)

    from trafficbots_amd.runtime import HipEngine, scene_from_batch

    cfg = load_model_config(overrides={"time_step_end": STEP_END, "n_joint_future": 1, "operand_precision": args.operand_precision})
    sd = synth.make_state_dict(7)
    eng = HipEngine(cfg, f"cuda:{local_rank}")
    eng.load_state_dict(sd)
    # this rank's shard of the global batch: scenes [rank*32, rank*32+32) of the seeded stream (configs[2] layout)
    batch = synth.make_batch(5000, B_PER_GPU, scene_offset=rank * B_PER_GPU, n_agent=N_AGENT, n_pl=N_PL, n_tl=N_TL)
    scene = scene_from_batch(batch, dev)

    # ---- one-time encoders (timed separately)
    enc = eng.encode_scene(scene)
    torch.cuda.synchronize()
    enc_t = []
    for _ in range(3):
        t0 = time.time()
        enc = eng.encode_scene(scene)
        torch.cuda.synchronize()
        enc_t.append((time.time() - t0) * 1e3)
    encode_ms = sorted(enc_t)[1]  # median of three
    feats = {"map_feature": enc["map_feature"], "map_feature_valid": enc["map_feature_valid"], "tl_feature": enc["tl_feature"]}
    z = enc["latent_mean"].clone()  # deterministic personality (K = 1)
    dest = enc["dest_logits"].argmax(-1).to(torch.int32)
    gv = scene["agent_valid"].bool().any(1).to(torch.uint8)

    # a generation-2 collection of the interpreter (tens of ms with the weight / scene dicts alive) in the launching thread
    # starves the stream right after a synchronize, when nothing is queued ahead: collect now, keep the collector off while timing
    import gc

    gc.collect()
    gc.disable()
    # keys of the traffic-light attention the step kernel actually walks: valid stop points per (scene, history step), whole blocks
    tl_cnt = scene["tl_valid"].sum(-1).float()
    tl_keys_eff = float(torch.clamp(torch.ceil(tl_cnt / 32) * 32, min=32).mean())
    out = None
    for _ in range(args.warmup):
        out = eng.rollout(scene, feats, z, enc["latent_mean"], dest, gv, 1, STEP_END, out=out)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    pass_ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]  # one marker per pass, same stream
    t0 = time.perf_counter()
    pass_ev[0].record()
    for i in range(args.steps):
        if i == args.steps - 1:
            eng.set_timing(True)  # hipEventRecord markers around the per-step kernels of the LAST timed pass (no host sync)
        out = eng.rollout(scene, feats, z, enc["latent_mean"], dest, gv, 1, STEP_END, out=out)
        pass_ev[i + 1].record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    gc.enable()
    pass_ms = [pass_ev[i].elapsed_time(pass_ev[i + 1]) for i in range(args.steps)]
    if os.environ.get("TB_BENCH_VERBOSE"):
        print("pass_ms:", " ".join(f"{x:.2f}" for x in pass_ms), file=sys.stderr)
    pass_ms = sorted(pass_ms)
    tm = eng.get_timing()  # HIP-event durations of the LAST pass of the timed region
    k_us = tm["fused_ms"] / max(1, tm["n_fused"]) * 1e3  # average duration of one fused k_step launch (C(t)+A(t+1))

    # ---- secondary measurement (not `value`): two independent 32-scene batches in flight on two HIP streams.  At 32 scenes a
    # rollout launches 128 workgroups (one per 16 agents) on a 256-CU chip; a second batch on another stream fills the rest.
    two_stream = None
    if world == 1:
        eng2 = HipEngine(cfg, f"cuda:{local_rank}")
        eng2.load_state_dict(sd)
        s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
        out2 = None
        for e_, st_ in ((eng, s1), (eng2, s2)):  # warm-up of the second context / streams
            with torch.cuda.stream(st_):
                o_ = e_.rollout(scene, feats, z, enc["latent_mean"], dest, gv, 1, STEP_END, out=(out if e_ is eng else out2))
                if e_ is eng2:
                    out2 = o_
        torch.cuda.synchronize()
        n_pairs = max(1, args.steps // 2)
        t1 = time.perf_counter()
        for _ in range(n_pairs):
            with torch.cuda.stream(s1):
                eng.rollout(scene, feats, z, enc["latent_mean"], dest, gv, 1, STEP_END, out=out)
            with torch.cuda.stream(s2):
                eng2.rollout(scene, feats, z, enc["latent_mean"], dest, gv, 1, STEP_END, out=out2)
        torch.cuda.synchronize()
        dt2 = time.perf_counter() - t1
        two_stream = {"value": 2 * n_pairs * B_PER_GPU * STEP_END / dt2, "unit": "scene-steps/s", "passes": 2 * n_pairs,
                      "identical_results": bool(torch.equal(out["preds"], out2["preds"])),
                      "note": "two independent batches of 32 scenes overlapped on two streams (64 scenes in flight); not the headline value"}
        del eng2

    # ---- metric partials + the one collective of the path (torchmetrics dist_reduce_fx="sum" states in the reference)
    from trafficbots_amd.shard import PARTIAL_FIELDS, all_reduce_partials, metric_partials

    part = metric_partials(out["preds"], out["valid"], out["outside_map"], out["dest_reached"], B_PER_GPU, STEP_END)
    # the reference's own metric states for this buffer (TrafficRuleMetrics sums; no ground-truth future in the test split, so the
    # ErrorMetrics sums stay 0): tb_metric_partials, appended to the same all-reduced vector
    from trafficbots_amd.runtime import METRIC_FIELDS

    k1 = lambda x: x.reshape(B_PER_GPU, N_AGENT, 1, *x.shape[2:])  # noqa: E731  [N,A,S,..] -> [B,A,K=1,S,..]
    ref_part = eng.metric_partials(
        k1(out["valid"]), k1(out["preds"]), k1(out["override_masks"]),
        {"outside_map": k1(out["outside_map"]), "dest_reached": k1(out["dest_reached"])},
        scene["agent_type"], torch.from_numpy(batch["history/agent/role"]))
    red_all, elapsed = all_reduce_partials(torch.cat([part, ref_part]), elapsed, fields=PARTIAL_FIELDS + METRIC_FIELDS)
    red = {k: red_all[k] for k in PARTIAL_FIELDS}
    ref_metrics = {k: red_all[k] for k in METRIC_FIELDS}
    finite = bool(torch.isfinite(out["preds"]).all())

    if rank == 0:
        total_scene_steps = world * B_PER_GPU * STEP_END * args.steps
        value = total_scene_steps / elapsed
        n_inst = B_PER_GPU
        fl = (flops_step_a(N_AGENT, N_PL, N_TL) + flops_step_c(N_AGENT)) * n_inst  # 176.1 MFLOP per scene-step
        achieved = fl / (k_us * 1e-6) / 1e12
        line = {
            "metric": "rollout scene-steps/sec (64 agents, 90 executed = 10 teacher-forced + 80 free steps)",
            "value": value, "unit": "scene-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": ("f32 (fp16-pair operands on the XDL MFMA, fp32 accumulate; fp32 everywhere else)" if args.operand_precision == "fp32"
                      else "bf16 MFMA operands, fp32 accumulate and state (BASELINE configs 4/5 precision; NOT the headline fp32 metric)"),
            "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[1]: 32 synthetic WOMD-shaped scenes per GPU, 64 agents, 256 polylines, "
                                   "40 TL stop points, K=1, 90-step closed-loop rollout, fp32, random-init weights",
                       "scenes_per_gpu": B_PER_GPU, "n_agent": N_AGENT, "n_pl": N_PL, "n_tl": N_TL, "sim_steps": STEP_END,
                       "parallelism": f"scene-parallel x{world}"},
            "agent_steps_per_s": value * N_AGENT,
            "encode_ms": encode_ms,
            "pass_ms": {"min": pass_ms[0], "median": pass_ms[len(pass_ms) // 2], "max": pass_ms[-1]},
            "kernel_us": {"k_step_fused": k_us, "n_fused": tm["n_fused"], "edge_launches_ms": tm["edge_ms"], "prologue_ms": tm["prologue_ms"],
                          "note": "per rollout: n_fused fused launches; edge_launches_ms = the batched warm-start launch (A halves of the "
                                  "teacher-forced steps, k_step_x<true>), their C-only launches and the last step's"},
            "roofline": {"bound": "mfma", "kernel": "tb::xh::k_step_x<false> (the fused C(t)+A(t+1) launch of a simulation step)", "achieved": achieved,
                         "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_FP32_MFMA_TFLOPS,
                         "traffic": TRAFFIC_BYTES_PER_LAUNCH_B32 if (N_AGENT, N_PL, N_TL, B_PER_GPU) == (64, 256, 40, 32) else None,
                         "traffic_unit": "bytes per launch (rocprofv3 --pmc FETCH_SIZE, profiles/r01_rocprof_xdl.txt)",
                         "flops_per_launch": fl, "avg_launch_us": k_us,
                         "note": "algorithmic fp32 flops = 176.1 MFLOP per scene-step (SURVEY 8(d)) x 32 scenes per launch, priced against "
                                 "the fp32-MFMA peak of the fp32 formulation; the kernel issues them as 3 fp16 MFMAs per product on the XDL "
                                 "pipe (fp32-accurate, DESIGN.md 4) and is bound by the per-CU vector-load path, see load_path; 128 "
                                 "workgroups (one per 16 agents) occupy 128 of 256 CUs at this batch size",
                         "load_path": load_path(k_us, tl_keys_eff),
                         "hbm": {"achieved": TRAFFIC_BYTES_PER_LAUNCH_B32 / (k_us * 1e-6) / 1e9, "peak": 8000.0, "unit": "GB/s",
                                 "frac": TRAFFIC_BYTES_PER_LAUNCH_B32 / (k_us * 1e-6) / 1e9 / 8000.0,
                                 "note": "measured FETCH bytes per launch / launch time: the launch is not HBM-bound"}
                         if (N_AGENT, N_PL, N_TL, B_PER_GPU) == (64, 256, 40, 32) else None},
            "checks": dict(finite=finite, **red),
            "reference_metric_states": dict(ref_metrics, note="sum-states of the reference's TrafficRuleMetrics / ErrorMetrics over all ranks "
                                            "(tb_metric_partials + the all-reduce); e.g. dest_reached / counter_agent = "
                                            f"{ref_metrics['dest_reached'] / max(1.0, ref_metrics['counter_agent']):.4f}"),
            "two_batches_in_flight": two_stream,
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
            if line["cpu_baseline"]["value"]:
                line["gpu_over_cpu"] = value / line["cpu_baseline"]["value"]
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
