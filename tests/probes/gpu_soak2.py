"""Development probe (GPU box), second soak: the parts of the mirror the first one (gpu_soak.py) does not touch -- validation_step
(plain and through lanes: metric holders), training_step (forward value), iterators abandoned half way (pipeline / prefetch), weights
reloaded many times, a new random shape every batch (grow-only workspaces, shape-keyed caches), the packed-h5 loader feeding
validation steps.  Reports memory after each phase; exits non-zero on growth or on a result that changes between repetitions."""
import gc
import hashlib
import os
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from trafficbots_amd import synth  # noqa: E402
from trafficbots_amd.waymo_motion import WaymoMotion  # noqa: E402

REPS = int(os.environ.get("REPS", "6"))
sd = synth.make_state_dict(7)
fails = []


def mem():
    torch.cuda.synchronize()
    free, total = torch.cuda.mem_get_info()
    return torch.cuda.memory_allocated() / 2**20, (total - free) / 2**20


def h(*ts):
    d = hashlib.sha256()
    for t in ts:
        d.update(t.detach().contiguous().cpu().numpy().tobytes())
    return d.hexdigest()[:12]


def phase(name, fn, reps=REPS, tol_mb=48.0):
    """fn() -> digest string; run twice to warm, then `reps` times with the cyclic collector off"""
    d0 = fn()
    fn()
    gc.collect()
    gc.disable()
    a0, u0 = mem()
    t0 = time.perf_counter()
    try:
        for _ in range(reps):
            d = fn()
            if d != d0:
                fails.append(f"{name}: result changed between repetitions ({d} != {d0})")
    finally:
        gc.enable()
    a1, u1 = mem()
    dt = (time.perf_counter() - t0) / reps
    ok = (a1 - a0) <= 1.0 and (u1 - u0) <= tol_mb
    if not ok:
        fails.append(f"{name}: memory grew (torch allocated {a1 - a0:+.2f} MB, device {u1 - u0:+.1f} MB)")
    print(f"  {name:58s} {dt * 1e3:9.1f} ms per repetition   torch allocated {a0:8.1f} -> {a1:8.1f} MB   device used {u0:8.1f} -> {u1:8.1f} MB   {'ok' if ok else 'GROWTH'}", flush=True)


scene = dict(n_agent=24, n_pl=64, n_tl=12)
wm = WaymoMotion(time_step_end=50, n_joint_future=2)
wm.load_state_dict(sd)
tb = [synth.make_batch(7000 + i, 4, **scene) for i in range(4)]
vb = [synth.make_val_batch(7100 + i, 3, p_future_spawn=0.3, p_future_exit=0.3, **scene) for i in range(4)]
eps_t = torch.from_numpy(synth.make_latent_noise(3, 8, 24)).cuda()
eps_v = torch.from_numpy(synth.make_latent_noise(4, 6, 24)).cuda()
gen = lambda i: torch.Generator(device="cuda").manual_seed(50 + i)  # noqa: E731


def val_plain():
    outs = [wm.validation_step(b, latent_eps=eps_v, generator=gen(i)) for i, b in enumerate(vb)]
    return h(*[o["joint_future_pred"]["rollout_buffer"].preds for o in outs], *[o["reactive_replay"]["metric_states"] for o in outs])


def val_lanes():
    outs = list(wm.pipeline(vb, lanes=2, step="validation_step", kwargs_fn=lambda i: dict(latent_eps=eps_v, generator=gen(i))))
    return h(*[o["joint_future_pred"]["rollout_buffer"].preds for o in outs], *[o["reactive_replay"]["metric_states"] for o in outs])


def train_plain():
    outs = [wm.training_step(b, latent_eps=eps_v[:3], generator=gen(i)) for i, b in enumerate(vb)]
    return h(*[torch.as_tensor(float(o["loss"]) if not torch.is_tensor(o["loss"]) else o["loss"]).reshape(1).cuda() for o in outs])


def abandoned_iterators():
    it = iter(wm.pipeline(tb * 2, lanes=2, kwargs_fn=lambda i: dict(latent_eps=eps_t, generator=gen(i))))
    a = next(it)
    b = next(it)
    d = h(a["rollout_buffer"].preds, b["rollout_buffer"].preds)
    it.close()  # the consumer leaves after two of eight batches: the lanes' work in flight is drained, nothing dangles
    pf = iter(wm.prefetch(tb * 2))
    o = wm.test_step(next(pf), latent_eps=eps_t, generator=gen(0))
    d2 = h(o["rollout_buffer"].preds)
    del pf
    return d + d2


def reload_weights():
    wm.load_state_dict(sd)
    o = wm.test_step(tb[0], latent_eps=eps_t, generator=gen(0))
    return h(o["rollout_buffer"].preds)


rng = np.random.default_rng(123)
shapes = [(int(rng.integers(1, 6)), int(rng.integers(1, 40)), int(rng.integers(2, 140)), int(rng.integers(1, 20))) for _ in range(10)]
rand_batches = [synth.make_batch(7300 + i, b, n_agent=a, n_pl=p, n_tl=t) for i, (b, a, p, t) in enumerate(shapes)]


def random_shapes():
    outs = [wm.test_step(b, generator=gen(i)) for i, b in enumerate(rand_batches)]
    return h(*[o["rollout_buffer"].preds for o in outs])


print(f"soak 2: {REPS} repetitions per phase after two warm-up calls, cyclic collector off inside a phase")
phase("validation_step x4, plain", val_plain)
phase("validation_step x4, wm.pipeline(lanes=2)", val_lanes)
phase("training_step x4 (forward value)", train_plain)
phase("pipeline / prefetch iterators abandoned half way", abandoned_iterators)
phase("load_state_dict + test_step", reload_weights, tol_mb=64.0)
phase("ten random shapes per repetition (test_step)", random_shapes, tol_mb=64.0)

if os.environ.get("H5", "1") == "1":
    try:
        from trafficbots_amd import data_h5

        tmp = tempfile.mkdtemp()
        episodes, attrs = synth.make_h5_episodes(7400, 12, n_agent=24, n_pl=64, n_tl=12)
        test_eps = [{k: v for k, v in e.items() if k.startswith(("history/", "map/"))} for e in episodes]
        data_h5.write_packed_h5(os.path.join(tmp, "testing.h5"), test_eps, attrs, deflate=0)
        dm = data_h5.DataH5womd(tmp, batch_size=4, n_agent=24, n_pl=64, n_tl_stop=12)
        for key in list(dm.tensor_size_test):
            dm.tensor_size_test[key] = test_eps[0][key].shape
        dm.setup("test")
        loader = dm.test_dataloader()

        def h5_test():
            outs = [wm.test_step(b, generator=gen(i)) for i, b in enumerate(loader)]
            return h(*[o["rollout_buffer"].preds for o in outs])

        def h5_test_lanes():
            outs = list(wm.pipeline(loader, lanes=2, kwargs_fn=lambda i: dict(generator=gen(i))))
            return h(*[o["rollout_buffer"].preds for o in outs])

        phase("packed-h5 loader -> test_step (3 batches of 4 episodes)", h5_test, tol_mb=64.0)
        phase("packed-h5 loader -> wm.pipeline(lanes=2)", h5_test_lanes, tol_mb=64.0)
    except Exception as e:
        import traceback

        traceback.print_exc()
        fails.append("packed-h5 phase: " + repr(e)[:200])
print("FAILURES:" if fails else "no growth, no changed result")
for f in fails:
    print("  ", f)
sys.exit(1 if fails else 0)
