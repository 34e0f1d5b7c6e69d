"""Host -> device staging of a batch for the hot path: ONE pinned buffer, ONE host-to-device copy, no device-side glue.

The reference hands `test_step` / `validation_step` a dict of ~30 host tensors (`src/data_modules/data_h5_womd.py:85-171`: bool
one-hots, [.., 1]-shaped yaw / speed columns) and converts them on the device inside `SceneCentricPreProcessing`
(`src/data_modules/scene_centric.py:92-133`).  Done literally that is ~20 pageable copies and ~40 small conversion kernels per
batch in front of the encoders (profiles/r06_e2e_before.txt).  Here the conversion to the C ABI's layout (uint8 masks, int32 class
indices, [B, S, A] columns, `agent_state` = [pos, yaw, spd]) happens ON THE HOST while the batch is copied into a pinned slab -- a
few hundred microseconds of numpy over ~3.5 MB at the headline shape -- and the slab crosses PCIe once; every scene tensor is a view
of the one device buffer.  What depends only on the host data and would otherwise cost device kernels travels with it: the default
teacher-forcing mask (`src/utils/teacher_forcing.py:33-74`), `goal_valid = agent_valid.any(1)`, the one-hot copies the harness reads
back as "ref/*" (`scene_centric.py:127-133`).

`BatchPrefetcher` overlaps batch n + 1's staging AND its scene encoders (a side stream) with batch n's rollout: `test_step` calls
back into it after it has enqueued the rollout and before it synchronises for the range check, so the host packing runs while the GPU
is busy and the encoders fill the CUs the 128-tile step launches leave idle.

No arithmetic of the path runs here (layout conversion only), and nothing falls back to the CPU: the consumers of a staged scene are
the HIP entry points.
"""
from __future__ import annotations

import weakref
from math import prod
from typing import Dict, Iterable, Iterator, List, Optional, Sequence, Tuple

import numpy as np
import torch
from torch import Tensor

_ALIGN = 256
_NP2TORCH = {np.dtype(np.float32): torch.float32, np.dtype(np.uint8): torch.uint8, np.dtype(np.int32): torch.int32,
             np.dtype(np.bool_): torch.bool}


def _np(x) -> np.ndarray:
    """Host array of a batch entry (numpy array or CPU tensor) without a copy."""
    if isinstance(x, np.ndarray):
        return x
    if torch.is_tensor(x):
        return x.detach().numpy()
    return np.asarray(x)


def is_host_batch(batch: Dict) -> bool:
    """True when every tensor of the batch lives on the host (numpy arrays / CPU tensors): the staged path applies."""
    return all(not (torch.is_tensor(v) and v.is_cuda) for v in batch.values())


def onehot_to_index_np(x: np.ndarray, out: Optional[np.ndarray] = None) -> np.ndarray:
    """bool one-hot [..., C] -> int32 index, the first set class, -1 where none is set (`runtime._onehot_to_index` on the host):
    `tb_host_onehot_index` of the HIP library (plain host code, GIL released; numpy reduces a short last axis slowly)."""
    from . import hip

    x = np.ascontiguousarray(x)
    if x.dtype != np.bool_ and x.dtype != np.uint8:
        x = x != 0
    if out is None:
        out = np.empty(x.shape[:-1], dtype=np.int32)
    assert out.flags.c_contiguous and out.dtype == np.int32 and out.shape == x.shape[:-1]
    hip.load().tb_host_onehot_index(x.ctypes.data, out.size, x.shape[-1], out.ctypes.data)
    return out


def teacher_forcing_mask_np(valid: np.ndarray, step_spawn_agent: int = 10, step_warm_start: int = 10) -> np.ndarray:
    """`TeacherForcing.get` with the schedule terms at their default 0 (`src/utils/teacher_forcing.py:33-74`;
    `runtime.teacher_forcing_mask` on the host).  valid [B, S, A] bool -> uint8 mask."""
    valid = valid.astype(bool, copy=False)
    m = np.zeros_like(valid)
    m[:, 0] |= valid[:, 0]
    if step_spawn_agent > 0:
        sp = (~valid[:, :-1]) & valid[:, 1:]
        sp[:, step_spawn_agent:] = False
        m[:, 1:] |= sp
    if step_warm_start >= 0:
        m[:, : step_warm_start + 1] |= valid[:, : step_warm_start + 1]
    return m.view(np.uint8)


def no_early_exit_np(valid: np.ndarray, n_steps: int) -> bool:
    v = valid[:, :n_steps].astype(bool, copy=False)
    return not bool((v[:, :-1] & ~v[:, 1:]).any())


class _Plan:
    """Byte layout of one slab: (name, numpy dtype, shape, offset) per field, every field 256-byte aligned."""

    def __init__(self) -> None:
        self.fields: List[Tuple[str, np.dtype, Tuple[int, ...], int]] = []
        self.nbytes = 0

    def add(self, name: str, dtype, shape: Sequence[int]) -> None:
        dt = np.dtype(dtype)
        shape = tuple(int(s) for s in shape)
        n = prod(shape) * dt.itemsize
        self.fields.append((name, dt, shape, self.nbytes))
        self.nbytes = (self.nbytes + n + _ALIGN - 1) // _ALIGN * _ALIGN

    def host_views(self, buf: np.ndarray) -> Dict[str, np.ndarray]:
        return {name: buf[off: off + prod(shape) * dt.itemsize].view(dt).reshape(shape) for name, dt, shape, off in self.fields}

    def device_views(self, dev: Tensor) -> Dict[str, Tensor]:
        out = {}
        for name, dt, shape, off in self.fields:
            out[name] = dev[off: off + prod(shape) * dt.itemsize].view(_NP2TORCH[dt]).view(shape)
        return out


def _agent_part(plan: _Plan, pre: str, b: int, s: int, a: int) -> None:
    f32, u8, i32 = np.float32, np.uint8, np.int32
    plan.add(pre + "agent_valid", u8, (b, s, a))
    plan.add(pre + "agent_pos", f32, (b, s, a, 2))
    plan.add(pre + "agent_yaw", f32, (b, s, a))
    plan.add(pre + "agent_spd", f32, (b, s, a))
    plan.add(pre + "agent_state", f32, (b, s, a, 4))
    plan.add(pre + "agent_vel", f32, (b, s, a, 2))
    plan.add(pre + "agent_acc", f32, (b, s, a))
    plan.add(pre + "agent_yaw_rate", f32, (b, s, a))
    plan.add(pre + "agent_type", i32, (b, a))
    plan.add(pre + "agent_size", f32, (b, a, 3))


def _tl_part(plan: _Plan, pre: str, b: int, s: int, t: int) -> None:
    plan.add(pre + "tl_valid", np.uint8, (b, s, t))
    plan.add(pre + "tl_state", np.int32, (b, s, t))
    plan.add(pre + "tl_pos", np.float32, (b, s, t, 2))
    plan.add(pre + "tl_dir", np.float32, (b, s, t, 2))


def _fill_agents(v: Dict[str, np.ndarray], pre: str, batch: Dict, key: str, steps: Optional[int]) -> None:
    def h(k):
        x = _np(batch[key + k])
        return x if steps is None else x[:, :steps]

    np.copyto(v[pre + "agent_valid"], h("agent/valid"), casting="unsafe")
    pos, yaw, spd = h("agent/pos"), h("agent/yaw_bbox")[..., 0], h("agent/spd")[..., 0]
    np.copyto(v[pre + "agent_pos"], pos, casting="unsafe")
    np.copyto(v[pre + "agent_yaw"], yaw, casting="unsafe")
    np.copyto(v[pre + "agent_spd"], spd, casting="unsafe")
    st = v[pre + "agent_state"]
    np.copyto(st[..., :2], pos, casting="unsafe")
    np.copyto(st[..., 2], yaw, casting="unsafe")
    np.copyto(st[..., 3], spd, casting="unsafe")
    np.copyto(v[pre + "agent_vel"], h("agent/vel"), casting="unsafe")
    np.copyto(v[pre + "agent_acc"], h("agent/acc")[..., 0], casting="unsafe")
    np.copyto(v[pre + "agent_yaw_rate"], h("agent/yaw_rate")[..., 0], casting="unsafe")
    onehot_to_index_np(_np(batch[key + "agent/type"]), v[pre + "agent_type"])
    np.copyto(v[pre + "agent_size"], _np(batch[key + "agent/size"]), casting="unsafe")


def _fill_tl(v: Dict[str, np.ndarray], pre: str, batch: Dict, key: str, steps: Optional[int]) -> None:
    def h(k):
        x = _np(batch[key + k])
        return x if steps is None else x[:, :steps]

    np.copyto(v[pre + "tl_valid"], h("tl_stop/valid"), casting="unsafe")
    onehot_to_index_np(h("tl_stop/state"), v[pre + "tl_state"])
    np.copyto(v[pre + "tl_pos"], h("tl_stop/pos"), casting="unsafe")
    np.copyto(v[pre + "tl_dir"], h("tl_stop/dir"), casting="unsafe")


class HostStager:
    """A ring of pinned slabs on one device.  `stage(batch)` packs a reference-layout host batch into the next slab (waiting, if the
    slab's previous upload is still in flight), enqueues ONE host-to-device copy on the CURRENT stream and returns the scene dict of
    `runtime.scene_from_batch` -- every tensor a view of one device buffer -- plus, for a validation / training batch, `scene["gt"]`
    as `runtime.gt_from_batch` builds it."""

    def __init__(self, device, n_hist: int = 11, tf_params: Tuple[int, int] = (10, 10), depth: int = 3) -> None:
        self.device = torch.device(device)
        self.n_hist = int(n_hist)
        self.tf_params = (int(tf_params[0]), int(tf_params[1]))
        self._slabs: List[Optional[Tensor]] = [None] * depth
        self._events: List[Optional[torch.cuda.Event]] = [None] * depth
        self._next = 0
        self._plans: Dict = {}
        self.n_uploads = 0      # host-to-device copies issued (one per staged batch)
        self.bytes_uploaded = 0

    # ---- layout -------------------------------------------------------------------------------------------------------
    def plan(self, batch: Dict) -> Tuple[_Plan, Dict]:
        pre = "history/" if "history/agent/valid" in batch else ""
        if not pre and "agent/valid" not in batch:
            raise KeyError("batch carries neither 'history/agent/*' nor 'agent/*'")
        nh = self.n_hist
        b, _, a = _np(batch[pre + "agent/valid"]).shape
        p, n_node = _np(batch["map/valid"]).shape[1:3]
        t = _np(batch[pre + "tl_stop/valid"]).shape[2]
        sig = (pre, b, a, p, n_node, t, tuple(batch["agent/valid"].shape) if "agent/valid" in batch else None,
               tuple(batch["tl_stop/valid"].shape) if "tl_stop/valid" in batch else None,
               tuple(batch["agent/role"].shape) if "agent/role" in batch else None,
               tuple(batch["agent/goal"].shape) if "agent/goal" in batch else None, tuple(batch["map/type"].shape))
        if sig in self._plans:
            return self._plans[sig]
        plan = _Plan()
        _agent_part(plan, "", b, nh, a)
        plan.add("map_valid", np.uint8, (b, p, n_node))
        plan.add("map_type", np.int32, (b, p))
        plan.add("map_pos", np.float32, (b, p, n_node, 2))
        plan.add("map_dir", np.float32, (b, p, n_node, 2))
        plan.add("map_boundary", np.float32, (b, 4))
        _tl_part(plan, "", b, nh, t)
        plan.add("goal_valid", np.uint8, (b, a))
        plan.add("tf_mask", np.uint8, (b, nh, a))
        plan.add("ref_agent_type", np.bool_, (b, a, 3))
        plan.add("ref_map_type", np.bool_, (b, p, _np(batch["map/type"]).shape[-1]))
        with_gt = "agent/valid" in batch
        if with_gt:
            s = _np(batch["agent/valid"]).shape[1]
            _agent_part(plan, "gt/", b, s, a)
            _tl_part(plan, "gt/", b, _np(batch["tl_stop/valid"]).shape[1], t)
            plan.add("gt/agent_role", np.uint8, _np(batch["agent/role"]).shape)
            plan.add("gt/gt_dest", np.int32, (b, a))
            if "agent/goal" in batch:
                plan.add("gt/gt_goal", np.float32, _np(batch["agent/goal"]).shape)
            plan.add("gt/tf_mask", np.uint8, (b, s, a))
        self._plans[sig] = (plan, {"pre": pre, "with_gt": with_gt})
        return self._plans[sig]

    def fill(self, views: Dict[str, np.ndarray], batch: Dict, info: Dict) -> Dict:
        """Layout conversion of one batch into the slab's fields.  Returns the host-side facts the harness wants as Python values."""
        pre, nh = info["pre"], self.n_hist
        _fill_agents(views, "", batch, pre, nh)
        np.copyto(views["map_valid"], _np(batch["map/valid"]), casting="unsafe")
        mt = _np(batch["map/type"])
        onehot_to_index_np(mt, views["map_type"])
        np.copyto(views["map_pos"], _np(batch["map/pos"]), casting="unsafe")
        np.copyto(views["map_dir"], _np(batch["map/dir"]), casting="unsafe")
        np.copyto(views["map_boundary"], _np(batch["map/boundary"]), casting="unsafe")
        _fill_tl(views, "", batch, pre, nh)
        valid = views["agent_valid"]
        views["goal_valid"][...] = valid.any(1)
        views["tf_mask"][...] = teacher_forcing_mask_np(valid, *self.tf_params)
        np.copyto(views["ref_agent_type"], _np(batch[pre + "agent/type"]), casting="unsafe")
        np.copyto(views["ref_map_type"], mt, casting="unsafe")
        facts = {"warm_ok": no_early_exit_np(_np(batch[pre + "agent/valid"]), nh)}
        if info["with_gt"]:
            _fill_agents(views, "gt/", batch, "", None)
            _fill_tl(views, "gt/", batch, "", None)
            np.copyto(views["gt/agent_role"], _np(batch["agent/role"]), casting="unsafe")
            np.copyto(views["gt/gt_dest"], _np(batch["agent/dest"]), casting="unsafe")
            if "gt/gt_goal" in views:
                np.copyto(views["gt/gt_goal"], _np(batch["agent/goal"]), casting="unsafe")
            views["gt/tf_mask"][...] = teacher_forcing_mask_np(views["gt/agent_valid"], *self.tf_params)
            facts["gt_warm_ok"] = no_early_exit_np(_np(batch["agent/valid"]), nh)
        return facts

    # ---- slabs --------------------------------------------------------------------------------------------------------
    def _slab(self, nbytes: int) -> Tuple[int, Tensor]:
        i = self._next
        self._next = (i + 1) % len(self._slabs)
        if self._events[i] is not None:
            self._events[i].synchronize()  # the slab's last upload has left the host buffer
            self._events[i] = None
        buf = self._slabs[i]
        if buf is None or buf.numel() < nbytes:
            pin = self.device.type == "cuda"
            buf = torch.empty(max(nbytes, 1), dtype=torch.uint8, pin_memory=pin)
            self._slabs[i] = buf
        return i, buf

    def stage(self, batch: Dict) -> Dict[str, Tensor]:
        plan, info = self.plan(batch)
        i, slab = self._slab(plan.nbytes)
        facts = self.fill(plan.host_views(slab.numpy()[: plan.nbytes]), batch, info)
        if self.device.type == "cuda":
            dev = torch.empty(plan.nbytes, dtype=torch.uint8, device=self.device)
            dev.copy_(slab[: plan.nbytes], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self._events[i] = ev
        else:  # (CPU: the layout tests; views of a private copy, the ring slab is reused)
            dev = slab[: plan.nbytes].clone()
        self.n_uploads += 1
        self.bytes_uploaded += plan.nbytes
        return scene_from_views(plan.device_views(dev), facts, self.tf_params)


def scene_from_views(v: Dict[str, Tensor], facts: Dict, tf_params: Tuple[int, int]) -> Dict[str, Tensor]:
    """Device views of a staged slab -> the scene dict (`runtime.scene_from_batch`'s keys; "gt" nested as `gt_from_batch`'s), with the
    host-made extras under underscore keys: `_goal_valid`, `_tf_mask` (+ `_tf_params`), `_ref_agent_type`, `_ref_map_type`."""
    scene: Dict[str, Tensor] = {k: t for k, t in v.items() if "/" not in k and k not in ("goal_valid", "tf_mask", "ref_agent_type", "ref_map_type")}
    scene["warm_ok"] = facts["warm_ok"]
    scene["_goal_valid"] = v["goal_valid"]
    scene["_ref_agent_type"], scene["_ref_map_type"] = v["ref_agent_type"], v["ref_map_type"]
    m = v["tf_mask"]
    m._tb_warm_start = (tf_params[1], m._version)
    scene["_tf_mask"], scene["_tf_params"] = m, tuple(tf_params)
    if "gt/agent_valid" in v:
        gt = {k[3:]: t for k, t in v.items() if k.startswith("gt/") and k != "gt/tf_mask"}
        gt["warm_ok"] = facts["gt_warm_ok"]
        m = v["gt/tf_mask"]
        m._tb_warm_start = (tf_params[1], m._version)
        gt["_tf_mask"], gt["_tf_params"] = m, tuple(tf_params)
        scene["gt"] = gt
    return scene


class StagedBatch(dict):
    """A pre-processed scene (what `WaymoMotion.pre_processing` returns) whose upload -- and, from a `BatchPrefetcher`, whose scene
    encoders -- were enqueued ahead of time on a side stream.  `ready` is the event the consuming stream waits for; `enc` the encoder
    outputs (`HipEngine.encode_scene`) or None; `host` the original host batch (bookkeeping keys: scenario_id, episode_idx ...)."""

    ready: Optional[torch.cuda.Event] = None
    enc: Optional[Dict[str, Tensor]] = None
    host: Optional[Dict] = None
    _prefetcher_ref = None  # weak: the prefetcher holds the batch it staged ahead (a cycle otherwise; tests/probes/gpu_soak2.py)

    @property
    def prefetcher(self) -> Optional["BatchPrefetcher"]:
        return self._prefetcher_ref() if self._prefetcher_ref is not None else None

    def wait(self, stream: Optional[torch.cuda.Stream] = None) -> None:
        """Make `stream` (default: the current one) wait for the staging work; the tensors allocated on the side stream are marked
        as used by it (caching-allocator bookkeeping)."""
        if self.ready is None:
            return
        stream = stream or torch.cuda.current_stream()
        stream.wait_event(self.ready)
        seen = set()

        def mark(x):
            if torch.is_tensor(x) and x.is_cuda and x.untyped_storage().data_ptr() not in seen:
                seen.add(x.untyped_storage().data_ptr())
                x.record_stream(stream)

        for x in self.values():
            if isinstance(x, dict):
                for y in x.values():
                    mark(y)
            else:
                mark(x)
        for y in (self.enc or {}).values():
            mark(y)
        self.ready = None


class BatchPrefetcher:
    """`for staged in wm.prefetch(loader): out = wm.test_step(staged)` -- batch n + 1 is staged (host packing, ONE upload) and encoded
    (`tb_encode_scene`) on a side stream while batch n's rollout runs.  The staging of the next batch is triggered from inside the
    harness step (`WaymoMotion._after_enqueue`): after the step has enqueued its rollout and before it synchronises, so the host-side
    packing overlaps GPU work without a second thread.  A step that never calls back (a caller driving the stages by hand) is served
    as well: the next batch is then staged when the iterator is advanced."""

    def __init__(self, wm, loader: Iterable[Dict], encode: bool = True) -> None:
        self.wm, self.loader, self.encode = wm, loader, encode
        # one side stream per `wm`, kept: every torch.cuda.Stream() takes the next of the runtime's pooled streams, each of which costs a
        # hardware queue (~5 MB of device memory) the first time it is used -- a prefetcher per epoch would walk through the pool
        if getattr(wm, "_prefetch_stream", None) is None:
            wm._prefetch_stream = torch.cuda.Stream(device=wm.device)
        self._stream = wm._prefetch_stream
        self._it: Optional[Iterator[Dict]] = None
        self._ahead: Optional[StagedBatch] = None
        self._done = False
        self.n_staged = 0

    def _stage_one(self) -> Optional[StagedBatch]:
        if self._done:
            return None
        try:
            batch = next(self._it)
        except StopIteration:
            self._done = True
            return None
        from .waymo_motion import retarget_stand_ins

        with torch.cuda.stream(self._stream):
            scene = self.wm.pre_processing(batch)
            sb = StagedBatch(scene)
            retarget_stand_ins(sb)  # (the attr / pe stand-ins of the scene refer to the dict they live in, weakly)
            sb.host = batch
            if self.encode:
                gt = sb.get("gt")
                sb.enc = self.wm.engine.encode_scene({k: v for k, v in sb.items() if k != "gt"})
                if gt is not None:
                    sb["gt"] = gt
            ev = torch.cuda.Event()
            ev.record()
            sb.ready = ev
        sb._prefetcher_ref = weakref.ref(self)
        self.n_staged += 1
        return sb

    def advance(self) -> None:
        """Stage the next batch now (idempotent until the iterator moves on)."""
        if self._ahead is None:
            self._ahead = self._stage_one()

    def __iter__(self) -> Iterator[StagedBatch]:
        self._it = iter(self.loader)
        self._done = False
        self._ahead = None
        self.advance()
        while self._ahead is not None:
            cur, self._ahead = self._ahead, None
            yield cur
            self.advance()  # (no-op when the step called back)

    def __len__(self) -> int:
        return len(self.loader)


def _record_stream(obj, stream, _depth: int = 0) -> None:
    """`record_stream(stream)` on every CUDA tensor reachable from a step's result (dicts, sequences, RolloutBuffer-like objects)."""
    if torch.is_tensor(obj):
        if obj.is_cuda:
            obj.record_stream(stream)
    elif isinstance(obj, dict):
        for v in obj.values():
            _record_stream(v, stream, _depth + 1)
    elif isinstance(obj, (list, tuple)):
        for v in obj:
            _record_stream(v, stream, _depth + 1)
    elif _depth < 4 and hasattr(obj, "__dict__") and not isinstance(obj, type):
        for v in vars(obj).values():
            _record_stream(v, stream, _depth + 1)


class LanePipeline:
    """`for out in wm.pipeline(loader): ...` (default: three lanes) -- the harness step (`test_step` by default) of consecutive batches on `lanes`
    independent contexts, each with its own stream: batch n + 1 is staged, encoded and rolled out on lane (n + 1) % lanes while batch n
    still runs on its lane.  At 32 scenes a rollout launch has 128 tiles for 256 CUs and is bound by per-CU latency chains, so two
    rollouts in flight nearly double the chip's throughput (bench.py `two_batches_in_flight`); the one-batch-at-a-time call sequence of
    the reference's harness (`src/pl_modules/waymo_motion.py:902-949`) cannot show that, this iterator can.

    Results come out IN ORDER and only after their lane's range check (`tb_check_status`): an fp16-pair overflow re-runs that batch
    on the exact-fp32 kernels before it is handed out, exactly as a plain `test_step` does (same draws, same `fallback_policy`: the
    lane goes back to the fast kernels afterwards unless overflows keep coming) -- nothing unchecked leaves the iterator.
    Bit-identical to plain calls (the lanes run the same kernels on the same inputs; the L2 warmers, prefetch hints without a
    numerical effect, switch themselves off while a second context is active).  `kwargs_fn(i)` supplies per-batch keyword arguments
    (`latent_eps`, `generator`, ...)."""

    def __init__(self, wm, loader: Iterable[Dict], lanes: int = 3, step: str = "test_step", kwargs_fn=None) -> None:
        assert lanes >= 1
        self.loader, self.step, self.kwargs_fn = loader, step, kwargs_fn
        self.wms, self.streams = wm._lanes(lanes)  # (the other lanes' contexts are made once per `wm` and kept)
        self.n_reruns = 0
        self.notes = []  # what each re-run's context did afterwards (WaymoMotion._after_fallback)

    def __len__(self) -> int:
        return len(self.loader)

    def __iter__(self):
        from collections import deque

        inflight = deque()
        it = enumerate(iter(self.loader))
        raw = [getattr(type(w), self.step).__wrapped__ for w in self.wms]  # (the step without its own synchronising range check)
        main = torch.cuda.current_stream(self.wms[0].device)

        def launch() -> bool:
            try:
                i, batch = next(it)
            except StopIteration:
                return False
            lane = i % len(self.wms)
            kw = self.kwargs_fn(i) if self.kwargs_fn else {}
            s = self.streams[lane]
            s.wait_stream(main)  # (whatever the caller queued before -- e.g. tensors in kw -- is visible to the lane)
            import copy

            # (a step that accumulates metric states -- validation_step -- may turn out invalid: the holders' states are replaced,
            # never edited in place, so a shallow copy taken now restores them; same contract as waymo_motion._range_fallback)
            snap = [copy.copy(h.__dict__) for h in self.wms[lane]._metric_holders()] if self.step != "test_step" else []
            # (and the random draws: a re-run sees the SAME numbers as the invalid run -- the state of the batch's generator, or of
            # torch's default generators, as it is now)
            gen = kw.get("generator")
            rng = ("gen", gen.get_state()) if gen is not None else \
                  ("default", torch.random.get_rng_state(), torch.cuda.get_rng_state(self.wms[lane].device))
            with torch.cuda.stream(s):
                out = raw[lane](self.wms[lane], batch, **kw)
                ev = torch.cuda.Event()
                ev.record(s)
            inflight.append((lane, batch, kw, out, ev, snap, rng))
            return True

        try:
            for _ in self.wms:
                if not launch():
                    break
            while inflight:
                lane, batch, kw, out, ev, snap, rng = inflight.popleft()
                ev.synchronize()
                w = self.wms[lane]
                if w.check_range:
                    with torch.cuda.stream(self.streams[lane]):
                        if w.engine.check_status(raise_on_range=False):  # overflowed: the lane's context is on the exact kernels now
                            self.n_reruns += 1
                            for h, d in zip(w._metric_holders(), snap):
                                h.__dict__.clear()
                                h.__dict__.update(d)
                            # the draws of the invalid run again; afterwards the default generators continue where the batches
                            # launched since have left them
                            if rng[0] == "gen":
                                kw["generator"].set_state(rng[1])
                            else:
                                later = (torch.random.get_rng_state(), torch.cuda.get_rng_state(w.device))
                                torch.random.set_rng_state(rng[1])
                                torch.cuda.set_rng_state(rng[2], w.device)
                            n_before = w.n_fallbacks
                            out = getattr(w, self.step)(batch, **kw)     # (the checked step: re-run + its own check)
                            if rng[0] == "default":
                                torch.random.set_rng_state(later[0])
                                torch.cuda.set_rng_state(later[1], w.device)
                            torch.cuda.current_stream().synchronize()
                            # fallback_policy: back to the fast kernels, or stay (the checked step has done it itself if the
                            # re-run overflowed in the other kernel family and ran a second time)
                            self.notes.append(w._after_fallback() if w.n_fallbacks == n_before else "re-run twice; see the warning")
                        else:
                            w._fallback_hist.append(0)
                main.wait_stream(self.streams[lane])  # the consumer's stream sees the finished results
                _record_stream(out, main)             # (they were allocated on the lane's stream: allocator bookkeeping)
                launch()  # refill the lane before handing the result out: the GPU stays fed while the consumer works
                yield out
        finally:
            for s in self.streams:
                s.synchronize()
            # the other lanes' metric holders (validation_step) flow into the first lane's: the caller reads ONE set of states
            for w in self.wms[1:]:
                for h0, hc in zip(self.wms[0]._metric_holders(), w._metric_holders()):
                    if hc.states is not None:
                        h0.update(hc.states)
                        hc.reset()
