"""Development probe (GPU box): soak run of the drop-in path -- many batches of alternating shapes through `wm.pipeline` and plain
`test_step` calls on long-lived contexts.  Checks what a serving / evaluation job needs and a unit test cannot show: device memory
does not grow (the library's workspaces are grow-only and shape-keyed, the stager's slab ring is bounded), every repetition of a batch
gives the SAME bytes (determinism across thousands of launches, graph replays and lane switches), throughput does not drift.

    N=1500 python tests/probes/gpu_soak.py      -> one summary block on stdout (profiles/r06_soak.txt)
"""
import hashlib
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from trafficbots_amd import synth  # noqa: E402
from trafficbots_amd.waymo_motion import WaymoMotion  # noqa: E402

N = int(os.environ.get("N", "1500"))
SHAPES = [  # (scenes, agents, polylines, tl, K, steps)
    (32, 64, 256, 40, 1, 90),
    (8, 20, 50, 12, 3, 40),
    (16, 64, 1024, 40, 6, 90),
]
sd = synth.make_state_dict(7)


def digest(out) -> str:
    b = out["rollout_buffer"]
    h = hashlib.sha256()
    for t in (b.preds, b.valid, b.action_log_probs, out["scores"], out["pred_dict"]["trajs"] if "trajs" in out["pred_dict"] else b.preds):
        h.update(t.detach().contiguous().cpu().numpy().tobytes())
    return h.hexdigest()[:16]


def census():
    import collections
    import gc

    gc.collect()
    c = collections.Counter()
    for o in gc.get_objects():
        try:
            if torch.is_tensor(o) and o.is_cuda:
                c[(tuple(o.shape), str(o.dtype))] += 1
        except Exception:
            pass
    return c


def mem():
    free, total = torch.cuda.mem_get_info()
    return {"torch_alloc_MB": torch.cuda.memory_allocated() / 2**20, "torch_reserved_MB": torch.cuda.memory_reserved() / 2**20,
            "device_used_MB": (total - free) / 2**20}


wms, data, eps = [], [], []
for (b, a, p, t, k, s) in SHAPES:
    w = WaymoMotion(time_step_end=s, n_joint_future=k)
    w.load_state_dict(sd)
    wms.append(w)
    data.append([synth.make_batch(6000 + 17 * i + a, b, n_agent=a, n_pl=p, n_tl=t) for i in range(3)])
    eps.append(torch.from_numpy(synth.make_latent_noise(9, b * k, a)).cuda())

if os.environ.get("TRACE"):
    torch.cuda.memory._record_memory_history(max_entries=400000)
ref = {}
t_log, m_log = [], []
done = 0
rounds = 0
t_start = time.perf_counter()
while done < N:
    for si, w in enumerate(wms):
        k = SHAPES[si][4]
        kw = lambda i, si=si, k=k: dict(latent_eps=eps[si], generator=torch.Generator(device="cuda").manual_seed(1000 + (i % 3)))  # noqa: E731
        stream = [data[si][i % 3] for i in range(12)]
        t0 = time.perf_counter()
        if rounds % 2 == 0:
            outs = list(w.pipeline(stream, lanes=2, kwargs_fn=kw))
        else:
            outs = [w.test_step(bt, **kw(i)) for i, bt in enumerate(stream)]
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / len(stream)
        for i, o in enumerate(outs):
            key = (si, i % 3)
            d = digest(o)
            if key not in ref:
                ref[key] = d
            elif ref[key] != d:
                print(f"MISMATCH shape {si} batch {i % 3} after {done} batches: {d} != {ref[key]}", flush=True)
                sys.exit(1)
        done += len(stream)
        t_log.append((si, rounds % 2, dt))
        del outs
        if os.environ.get("TRACE") and rounds < 8:
            ms = torch.cuda.memory_stats()
            print(f"    round {rounds} shape {si} mode {rounds % 2}: allocated {ms['allocated_bytes.all.current'] / 2**20:8.1f} active {ms['active_bytes.all.current'] / 2**20:8.1f} MB", flush=True)
    rounds += 1
    if rounds == 2:
        c_early = census()
    if rounds in (1, 2) or rounds % 8 == 0:
        m_log.append((done, mem()))
wall = time.perf_counter() - t_start
m_log.append((done, mem()))
print(f"soak: {done} batches ({rounds} rounds x {len(SHAPES)} shapes x 12, alternating wm.pipeline(lanes=2) / plain test_step) in {wall:.1f} s; "
      f"every repetition of each of the {len(ref)} (shape, batch) pairs bit-identical to its first result")
for si, sh in enumerate(SHAPES):
    for mode, name in ((0, "2 lanes"), (1, "plain")):
        ts = [t for s_, m_, t in t_log if s_ == si and m_ == mode]
        if len(ts) >= 4:
            q = max(1, len(ts) // 4)
            print(f"  shape B={sh[0]} A={sh[1]} P={sh[2]} K={sh[4]} S={sh[5]} {name:8s}: first quarter {1e3 * sum(ts[:q]) / q:7.2f} ms per batch, "
                  f"last quarter {1e3 * sum(ts[-q:]) / q:7.2f} ms ({len(ts)} rounds)")
print("  memory (MB) after n batches:")
for n, m in m_log:
    print(f"    {n:6d}: torch allocated {m['torch_alloc_MB']:9.1f}  torch reserved {m['torch_reserved_MB']:9.1f}  device used {m['device_used_MB']:9.1f}")
if os.environ.get("TRACE"):
    import collections

    snap = torch.cuda.memory._snapshot()
    by, sz = collections.Counter(), collections.Counter()
    for seg in snap["segments"]:
        for blk in seg["blocks"]:
            if blk["state"] != "active_allocated":
                continue
            fr = [f for f in (blk.get("frames") or []) if "/repo/" in f["filename"]]
            key = " <- ".join(f"{os.path.basename(f['filename'])}:{f['line']}" for f in fr[:4]) or "?"
            by[key] += 1
            sz[key] += blk["size"]
    print("  live blocks by allocation site:")
    for k_, v_ in sz.most_common(14):
        print(f"   {v_ / 2**20:8.2f} MB in {by[k_]:4d} blocks  {k_}")
grown = census() - c_early
if grown:
    print("  live CUDA tensors that were not there after round 2:")
    for k_, v_ in grown.most_common(10):
        print(f"    +{v_}  {k_}")
g0, g1 = m_log[1][1]["device_used_MB"], m_log[-1][1]["device_used_MB"]
print(f"  device memory growth after the second round: {g1 - g0:+.1f} MB")
sys.exit(0 if g1 - g0 < 64.0 else 2)
