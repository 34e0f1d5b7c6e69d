"""Packed-h5 scene loader: host mirror of `src/data_modules/data_h5_womd.py` over `libtrafficbots_h5.so`
(C ABI: include/trafficbots_h5.h).

`DataH5womd` keeps the reference's constructor, `tensor_size_*` tables, `setup(stage)` and `*_dataloader()` names.  What a
loader yields differs in one respect: instead of one numpy dict per sample collated by worker processes (bool one-hot tensors that
`SceneCentricPreProcessing` converts on the device every step), a reader thread decodes a whole batch straight into pinned host
buffers in the layout the HIP entry points take -- uint8 masks, int32 class indices, [B, S, A] yaw / speed / acceleration -- under
"packed/<name>" keys (history / scene part) and "packed/gt/<name>" keys (91-step ground truth of validation and training files),
next to the reference's bookkeeping keys ("episode_idx", "scenario_id", "scenario_center", "scenario_yaw", "with_map").
`WaymoMotion.pre_processing` recognises such a batch and only uploads it.  Only the tensors the hot path consumes are read
(`PACKED_SCENE`, `PACKED_GT`); `read_reference_batch` returns the reference's full, un-decoded batch for any key table.

There is no fallback reader: without the built library (``__graft_entry__.build()``) or the HDF5 runtime every entry point raises.
"""
from __future__ import annotations

import ctypes as C
import os
import queue
import threading
from typing import Dict, Iterator, List, Optional, Sequence, Tuple

import numpy as np
import torch

F32, MASK_U8, ONEHOT_I32, I64 = 0, 1, 2, 3
_LIB_PATH = os.environ.get("TB_H5_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libtrafficbots_h5.so")
_HDF5_CANDIDATES = ("/opt/conda/lib/libhdf5.so.103", "libhdf5.so.103", "libhdf5_serial.so.103")
EXPORTS = ("tb_h5_last_error", "tb_h5_open", "tb_h5_close", "tb_h5_len", "tb_h5_episode_attrs", "tb_h5_batch_attrs", "tb_h5_dataset_shape", "tb_h5_read_key",
           "tb_h5_read_batch", "tb_h5_set_index_cache", "tb_h5_save_index", "tb_h5_load_index",
           "tb_h5_writer_open", "tb_h5_writer_options", "tb_h5_writer_episode", "tb_h5_writer_dataset", "tb_h5_writer_close")
_lib = None
N_THREADS = int(os.environ.get("TB_H5_THREADS", min(16, os.cpu_count() or 1)))  # decode workers of one batch read


class TbH5KeySpec(C.Structure):
    _fields_ = [("key", C.c_char_p), ("dims", C.POINTER(C.c_int64)), ("rank", C.c_int32), ("n_lead", C.c_int32), ("kind", C.c_int32),
                ("dummy_on_mismatch", C.c_int32), ("out", C.c_void_p)]


def load() -> C.CDLL:
    """dlopen the HDF5 runtime (TB_HDF5_LIB or the image's /opt/conda copy), then the reader library."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise RuntimeError(f"trafficbots_amd: h5 reader not built ({_LIB_PATH} missing); build it with __graft_entry__.build()")
    err = None
    for cand in ((os.environ["TB_HDF5_LIB"],) if os.environ.get("TB_HDF5_LIB") else _HDF5_CANDIDATES):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
            err = None
            break
        except OSError as e:
            err = e
    if err is not None:
        raise RuntimeError(f"trafficbots_amd: no HDF5 1.10 runtime found (set TB_HDF5_LIB): {err}")
    lib = C.CDLL(_LIB_PATH)
    i64p = C.POINTER(C.c_int64)
    lib.tb_h5_last_error.restype = C.c_char_p
    lib.tb_h5_open.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
    lib.tb_h5_close.argtypes = [C.c_void_p]
    lib.tb_h5_close.restype = None
    lib.tb_h5_len.argtypes = [C.c_void_p]
    lib.tb_h5_len.restype = C.c_int64
    lib.tb_h5_episode_attrs.argtypes = [C.c_void_p, C.c_int64, C.c_char_p, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_int32),
                                        C.POINTER(C.c_double), C.POINTER(C.c_int32)]
    lib.tb_h5_batch_attrs.argtypes = [C.c_void_p, i64p, C.c_int32, C.c_char_p, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_int32),
                                      C.POINTER(C.c_double), C.POINTER(C.c_int32)]
    lib.tb_h5_dataset_shape.argtypes = [C.c_void_p, C.c_int64, C.c_char_p, C.POINTER(C.c_int32), i64p, C.POINTER(C.c_int32)]
    lib.tb_h5_read_key.argtypes = [C.c_void_p, i64p, C.c_int32, C.c_char_p, i64p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
    lib.tb_h5_read_batch.argtypes = [C.c_void_p, i64p, C.c_int32, C.POINTER(TbH5KeySpec), C.c_int32, C.c_int32]
    lib.tb_h5_set_index_cache.argtypes = [C.c_void_p, C.c_int64]
    lib.tb_h5_save_index.argtypes = [C.c_void_p, C.c_char_p, C.c_int32]
    lib.tb_h5_load_index.argtypes = [C.c_void_p, C.c_char_p]
    lib.tb_h5_writer_open.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
    lib.tb_h5_writer_options.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32]
    lib.tb_h5_writer_episode.argtypes = [C.c_void_p, C.c_int64, C.c_char_p, C.POINTER(C.c_double), C.c_int32, C.c_double, C.c_int32]
    lib.tb_h5_writer_dataset.argtypes = [C.c_void_p, C.c_char_p, C.c_int32, i64p, C.c_int32, C.c_void_p]
    lib.tb_h5_writer_close.argtypes = [C.c_void_p, C.c_int64]
    _lib = lib
    return lib


def _check(rc: int, what: str) -> None:
    if rc != 0:
        raise RuntimeError(f"{what} failed ({rc}): {load().tb_h5_last_error().decode()}")


def reference_kind(key: str) -> int:
    """Storage class of a reference tensor by its key (`data_h5_womd.py:85-170` comments): bool, int64 or float32."""
    leaf = key.rsplit("/", 1)[-1]
    if leaf in ("valid", "type", "state", "role", "cmd"):
        return MASK_U8
    if leaf in ("object_id", "dest", "idx"):
        return I64
    return F32


_TORCH_DTYPE = {F32: torch.float32, MASK_U8: torch.uint8, ONEHOT_I32: torch.int32, I64: torch.int64}
_NP_DTYPE = {F32: np.float32, MASK_U8: np.uint8, ONEHOT_I32: np.int32, I64: np.int64}

# packed name -> (key below the prefix, kind, drop the trailing singleton dim)
PACKED_AGENT = {
    "agent_valid": ("agent/valid", MASK_U8, False), "agent_pos": ("agent/pos", F32, False), "agent_yaw": ("agent/yaw_bbox", F32, True),
    "agent_spd": ("agent/spd", F32, True), "agent_vel": ("agent/vel", F32, False), "agent_acc": ("agent/acc", F32, True),
    "agent_yaw_rate": ("agent/yaw_rate", F32, True), "agent_type": ("agent/type", ONEHOT_I32, False), "agent_size": ("agent/size", F32, False),
}
PACKED_TL = {"tl_valid": ("tl_stop/valid", MASK_U8, False), "tl_state": ("tl_stop/state", ONEHOT_I32, False),
             "tl_pos": ("tl_stop/pos", F32, False), "tl_dir": ("tl_stop/dir", F32, False)}
PACKED_MAP = {"map_valid": ("map/valid", MASK_U8, False), "map_type": ("map/type", ONEHOT_I32, False), "map_pos": ("map/pos", F32, False),
              "map_dir": ("map/dir", F32, False), "map_boundary": ("map/boundary", F32, False)}
PACKED_GT_ONLY = {"agent_role": ("agent/role", MASK_U8, False), "gt_dest": ("agent/dest", I64, False), "gt_goal": ("agent/goal", F32, False)}
_STEP_KEYS = ("valid", "pos", "z", "vel", "spd", "acc", "yaw_bbox", "yaw_rate", "state", "dir", "idx")  # tensors with a leading step dim


class PackedH5File:
    """One open packed file (`DatasetBase`, `data_h5_womd.py:9-18`).  Not thread-safe: one instance per reading thread."""

    def __init__(self, filepath: str) -> None:
        self.lib = load()
        self.filepath = filepath
        self._h = C.c_void_p()
        _check(self.lib.tb_h5_open(filepath.encode(), C.byref(self._h)), f"tb_h5_open({filepath})")
        self.dataset_len = int(self.lib.tb_h5_len(self._h))

    def __len__(self) -> int:
        return self.dataset_len

    def close(self) -> None:
        if self._h:
            self.lib.tb_h5_close(self._h)
            self._h = C.c_void_p()

    def __del__(self) -> None:
        try:
            self.close()
        except Exception:
            pass

    def set_index_cache(self, max_entries: int) -> None:
        _check(self.lib.tb_h5_set_index_cache(self._h, max_entries), "tb_h5_set_index_cache")

    def save_index(self, path: str, merge_existing: bool = True) -> None:
        _check(self.lib.tb_h5_save_index(self._h, path.encode(), int(merge_existing)), f"tb_h5_save_index({path})")

    def load_index(self, path: str) -> bool:
        """True when the index file existed, matched this data file and was loaded."""
        return self.lib.tb_h5_load_index(self._h, path.encode()) == 0

    def episode_attrs(self, episode: int) -> Dict:
        sid = C.create_string_buffer(256)
        center = (C.c_double * 3)()
        n_center, with_map, yaw = C.c_int32(), C.c_int32(), C.c_double()
        _check(self.lib.tb_h5_episode_attrs(self._h, episode, sid, 256, center, C.byref(n_center), C.byref(yaw), C.byref(with_map)),
               f"tb_h5_episode_attrs({episode})")
        return {"scenario_id": sid.value.decode(), "scenario_center": np.array(center[:n_center.value]), "scenario_yaw": float(yaw.value),
                "with_map": bool(with_map.value)}

    def dataset_shape(self, episode: int, key: str) -> Tuple[int, ...]:
        rank, dims, esz = C.c_int32(), (C.c_int64 * 8)(), C.c_int32()
        _check(self.lib.tb_h5_dataset_shape(self._h, episode, key.encode(), C.byref(rank), dims, C.byref(esz)), f"tb_h5_dataset_shape({key})")
        return tuple(dims[: rank.value])

    @staticmethod
    def decoded_shape(n_episode: int, size: Sequence[int], kind: int, n_lead: int) -> List[int]:
        """Shape of what tb_h5_read_batch writes for one key (before the optional squeeze of a trailing singleton)."""
        shape = list(size)
        if n_lead:
            shape[0] = n_lead
        if kind == ONEHOT_I32:
            shape = shape[:-1]
        return [n_episode] + shape

    def read_keys(self, episodes: Sequence[int], specs: Sequence[Tuple], pin: bool = False, n_threads: Optional[int] = None,
                  into: Optional[Sequence[torch.Tensor]] = None) -> List[torch.Tensor]:
        """One tb_h5_read_batch call.  specs: (key, stored size, kind, n_lead, squeeze) per tensor; returns the [len(episodes), *decoded
        shape] host tensors (pinned on request) in the same order.  `into`: decode straight into these caller-made contiguous tensors
        (views of one pinned slab: `read_packed_batch`) instead of allocating one per key."""
        outs, keep = [], []
        arr = (TbH5KeySpec * max(len(specs), 1))()
        for i, (key, size, kind, n_lead, squeeze) in enumerate(specs):
            shape = self.decoded_shape(len(episodes), size, kind, n_lead)
            if into is not None:
                out = into[i]
                assert out.is_contiguous() and out.dtype == _TORCH_DTYPE[kind] and out.numel() == int(np.prod(shape)), key
                out = out.view(shape)
            else:
                out = torch.empty(shape, dtype=_TORCH_DTYPE[kind], pin_memory=pin)
            dims = (C.c_int64 * max(len(size), 1))(*size)
            kb = key.encode()
            keep += [dims, kb]
            arr[i] = TbH5KeySpec(kb, dims, len(size), n_lead, kind, int("agent" in key), out.data_ptr())
            if squeeze:
                assert out.shape[-1] == 1
                out = out[..., 0]
            outs.append(out)
        ep = (C.c_int64 * max(len(episodes), 1))(*[int(e) for e in episodes])
        _check(self.lib.tb_h5_read_batch(self._h, ep, len(episodes), arr, len(specs), N_THREADS if n_threads is None else n_threads),
               f"tb_h5_read_batch({self.filepath})")
        return outs

    def read_key(self, episodes: Sequence[int], key: str, size: Tuple[int, ...], kind: int, n_lead: int = 0, squeeze: bool = False,
                 pin: bool = False, n_threads: Optional[int] = None) -> torch.Tensor:
        return self.read_keys(episodes, [(key, size, kind, n_lead, squeeze)], pin, n_threads)[0]

    def read_reference_batch(self, episodes: Sequence[int], tensor_size: Dict[str, Tuple[int, ...]], with_attrs: bool) -> Dict:
        """The collated batch `DataLoader(DatasetVal | DatasetTrain)` yields (`data_h5_womd.py:27-55`): bool / float32 / int64
        tensors of the stored shapes, un-decoded.  For parity tests and for callers that want the reference's own layout."""
        out: Dict = {"episode_idx": torch.tensor([int(e) for e in episodes], dtype=torch.int64)}
        if with_attrs:
            out.update(self._collate_attrs(episodes))
        keys = list(tensor_size)
        for k, t in zip(keys, self.read_keys(episodes, [(k, tensor_size[k], reference_kind(k), 0, False) for k in keys])):
            out[k] = t.view(torch.bool) if t.dtype == torch.uint8 else t
        return out

    def _collate_attrs(self, episodes: Sequence[int]) -> Dict:
        """The four episode attributes of a batch in one C call (a reader thread re-acquires the GIL once, not once per episode)."""
        n, cap = len(episodes), 256
        ep = (C.c_int64 * max(n, 1))(*[int(e) for e in episodes])
        ids = C.create_string_buffer(max(n, 1) * cap)
        centers, yaws = np.zeros((n, 3), np.float64), np.zeros((n,), np.float64)
        n_center, with_map = np.zeros((n,), np.int32), np.zeros((n,), np.int32)
        dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int32)
        _check(self.lib.tb_h5_batch_attrs(self._h, ep, n, ids, cap, centers.ctypes.data_as(dp), n_center.ctypes.data_as(ip),
                                          yaws.ctypes.data_as(dp), with_map.ctypes.data_as(ip)), "tb_h5_batch_attrs")
        nc = int(n_center.max()) if n else 2
        return {"scenario_id": [ids.raw[i * cap:(i + 1) * cap].split(b"\0", 1)[0].decode() for i in range(n)],
                "scenario_center": torch.from_numpy(centers[:, :nc].copy()), "scenario_yaw": torch.from_numpy(yaws),
                "with_map": torch.from_numpy(with_map.astype(np.bool_))}

    def read_packed_batch(self, episodes: Sequence[int], tensor_size: Dict[str, Tuple[int, ...]], split: str, n_hist: int = 11,
                          pin: bool = False, tf_params: Tuple[int, int] = (10, 10)) -> Dict:
        """split "test": scene from "history/*"; "val": scene from "history/*" + ground truth from "agent/*", "tl_stop/*";
        "train": both from "agent/*", "tl_stop/*" (the history is their first `n_hist` steps, `scene_centric.py:103-133`).

        Round 6: the whole batch is decoded into ONE (pinned) slab in the layout of `staging.HostStager` -- every "packed/<name>" entry
        is a view of it -- together with what depends only on host data (`agent_state`, `goal_valid`, the default teacher-forcing
        mask for `tf_params`, the one-hot copies behind "ref/*"): `scene_from_packed` then uploads the slab with one copy."""
        from . import staging

        out: Dict = {"episode_idx": torch.tensor([int(e) for e in episodes], dtype=torch.int64)}
        if split != "train":
            out.update(self._collate_attrs(episodes))
        pre, lead = ("", n_hist) if split == "train" else ("history/", 0)
        names, specs = [], []
        for name, (key, kind, squeeze) in {**PACKED_AGENT, **PACKED_TL}.items():
            stepped = key.rsplit("/", 1)[-1] in _STEP_KEYS
            names.append(name)
            specs.append((pre + key, tensor_size[pre + key], kind, lead if stepped else 0, squeeze))
        for name, (key, kind, squeeze) in PACKED_MAP.items():
            names.append(name)
            specs.append((key, tensor_size[key], kind, 0, squeeze))
        if split != "test":
            for name, (key, kind, squeeze) in {**PACKED_AGENT, **PACKED_TL, **PACKED_GT_ONLY}.items():
                names.append("gt/" + name)
                specs.append((key, tensor_size[key], kind, 0, squeeze))
        nb = len(episodes)
        plan = staging._Plan()
        shapes = {}
        for name, (key, size, kind, n_lead, squeeze) in zip(names, specs):
            shp = self.decoded_shape(nb, size, kind, n_lead)
            shapes[name] = shp[:-1] if squeeze else shp
            if kind != I64:  # (int64 ids are decoded beside the slab and narrowed into it)
                plan.add(name, _NP_DTYPE[kind], shapes[name])
        b, nh, a = shapes["agent_valid"]
        p, n_cls_map = shapes["map_type"][1], tensor_size["map/type"][-1]
        plan.add("agent_state", np.float32, (b, nh, a, 4))
        plan.add("goal_valid", np.uint8, (b, a))
        plan.add("tf_mask", np.uint8, (b, nh, a))
        plan.add("ref_agent_type", np.bool_, (b, a, 3))
        plan.add("ref_map_type", np.bool_, (b, p, n_cls_map))
        if split != "test":
            s_gt = shapes["gt/agent_valid"][1]
            plan.add("gt/agent_state", np.float32, (b, s_gt, a, 4))
            plan.add("gt/gt_dest", np.int32, (b, a))
            plan.add("gt/tf_mask", np.uint8, (b, s_gt, a))
        slab = torch.empty(max(plan.nbytes, 1), dtype=torch.uint8, pin_memory=pin)
        hv = plan.host_views(slab.numpy()[: plan.nbytes])
        tv = plan.device_views(slab)  # (the same fields as torch views of the slab: what the batch hands out)
        side = {}
        into = []
        for name, (key, size, kind, n_lead, squeeze) in zip(names, specs):
            if kind == I64:
                side[name] = torch.empty(shapes[name], dtype=torch.int64)
                into.append(side[name])
            else:
                into.append(tv[name])
        self.read_keys(episodes, specs, into=into)
        # ---- what depends only on the decoded host data
        def derive(pre_: str) -> None:
            st = hv[pre_ + "agent_state"]
            st[..., :2], st[..., 2], st[..., 3] = hv[pre_ + "agent_pos"], hv[pre_ + "agent_yaw"], hv[pre_ + "agent_spd"]
            hv[pre_ + "tf_mask"][...] = staging.teacher_forcing_mask_np(hv[pre_ + "agent_valid"], *tf_params)

        derive("")
        hv["goal_valid"][...] = hv["agent_valid"].any(1)
        hv["ref_agent_type"][...] = hv["agent_type"][..., None] == np.arange(3, dtype=np.int32)
        hv["ref_map_type"][...] = hv["map_type"][..., None] == np.arange(n_cls_map, dtype=np.int32)
        facts = {"warm_ok": staging.no_early_exit_np(hv["agent_valid"], n_hist)}
        if split != "test":
            derive("gt/")
            hv["gt/gt_dest"][...] = side["gt/gt_dest"].numpy()
            facts["gt_warm_ok"] = staging.no_early_exit_np(hv["gt/agent_valid"], n_hist)
        for name, t in tv.items():
            out["packed/" + name] = t
        out["packed/_slab"], out["packed/_plan"], out["packed/_facts"], out["packed/_tf_params"] = slab, plan, facts, tuple(tf_params)
        return out


def _early_exit_free(valid_u8: torch.Tensor, n_steps: int) -> bool:
    v = valid_u8[:, :n_steps].bool()
    return not bool((v[:, :-1] & ~v[:, 1:]).any())


def scene_from_packed(batch: Dict, device, n_hist: int = 11, tf_params: Tuple[int, int] = (10, 10)) -> Dict:
    """Packed loader batch -> the dict `runtime.scene_from_batch` builds (plus `scene["gt"]` as `runtime.gt_from_batch` builds it
    when the batch carries ground truth): ONE upload of the batch's slab, every tensor a view of the device copy."""
    from . import staging

    slab, plan = batch["packed/_slab"], batch["packed/_plan"]
    device = torch.device(device)
    if device.type == "cuda":
        dev = torch.empty(plan.nbytes, dtype=torch.uint8, device=device)
        dev.copy_(slab[: plan.nbytes], non_blocking=True)
    else:
        dev = slab[: plan.nbytes].clone()
    views = plan.device_views(dev)
    made_for = batch["packed/_tf_params"]
    scene = staging.scene_from_views(views, batch["packed/_facts"], made_for)
    if tuple(tf_params) != tuple(made_for):  # (the consumer asks for other teacher-forcing parameters: it makes its own mask)
        for d in (scene, scene.get("gt", {})):
            d.pop("_tf_mask", None)
            d.pop("_tf_params", None)
    return scene


class PackedSceneLoader:
    """Iterable over packed batches of one file.  `readers` reader threads (each with its own file handle, batch i goes to reader
    i % readers in every epoch, so a handle's chunk index sees the same episodes again) keep `prefetch` decoded batches each ahead of
    the consumer, which takes them in order; the C calls release the GIL, and while one reader walks HDF5 metadata (the library
    serialises that) the other decodes, allocates and enqueues.  `rank` / `world_size` deal episodes round-robin like the
    DistributedSampler of the reference's DDP run (`run.py:51-53`), without its padding: the last batches of a rank may simply be
    absent."""

    def __init__(self, filepath: str, tensor_size: Dict[str, Tuple[int, ...]], split: str, batch_size: int, n_hist: int = 11,
                 prefetch: int = 2, pin: Optional[bool] = None, rank: int = 0, world_size: int = 1, seed: int = 0,
                 limit_batches: Optional[int] = None, readers: int = 2, index_path: Optional[str] = None,
                 tf_params: Tuple[int, int] = (10, 10)) -> None:
        assert split in ("train", "val", "test") and readers >= 1
        self.filepath, self.tensor_size, self.split, self.batch_size, self.n_hist = filepath, tensor_size, split, batch_size, n_hist
        self.prefetch, self.pin = prefetch, torch.cuda.is_available() if pin is None else pin
        self.rank, self.world_size, self.seed, self.limit_batches = rank, world_size, seed, limit_batches
        self.tf_params = tuple(tf_params)  # the default teacher-forcing mask travels with the batch (teacher_forcing_joint_future_pred)
        # kept across epochs: a handle's chunk index makes every later visit of an episode metadata-free
        self._files = [PackedH5File(filepath) for _ in range(readers)]
        self._busy = threading.Lock()  # one iteration at a time uses them; a concurrent second one opens its own
        self.dataset_len = len(self._files[0])
        self._epoch = 0
        # chunk index on disk: loaded now when it matches the data file, (re)written after the first complete epoch
        self.index_path = index_path
        self._index_saved = False
        if index_path and os.path.exists(index_path):
            self._index_saved = all([f.load_index(index_path) for f in self._files]) and split != "train"

    def _indices(self) -> List[int]:
        if self.split == "train":  # DatasetTrain.__getitem__ draws a random episode per item (:31)
            rng = np.random.default_rng([self.seed, self._epoch, self.rank])
            n = len(range(self.rank, self.dataset_len, self.world_size))
            return [int(i) for i in rng.integers(0, self.dataset_len, n)]
        return list(range(self.rank, self.dataset_len, self.world_size))

    def __len__(self) -> int:
        n = -(-len(range(self.rank, self.dataset_len, self.world_size)) // self.batch_size)
        return n if self.limit_batches is None else min(n, self.limit_batches)

    def __iter__(self) -> Iterator[Dict]:
        idx = self._indices()
        self._epoch += 1
        chunks = [idx[i:i + self.batch_size] for i in range(0, len(idx), self.batch_size)][: len(self)]
        mine = self._busy.acquire(blocking=False)
        files = self._files if mine else [PackedH5File(self.filepath) for _ in self._files]
        n = len(files)
        queues = [queue.Queue(maxsize=max(self.prefetch, 1)) for _ in range(n)]
        stop = threading.Event()

        def work(r: int) -> None:
            try:
                for c in chunks[r::n]:
                    if stop.is_set():
                        break
                    queues[r].put(files[r].read_packed_batch(c, self.tensor_size, self.split, self.n_hist, self.pin, self.tf_params))
            except BaseException as e:  # surfaced in the consumer, in order
                queues[r].put(e)

        threads = [threading.Thread(target=work, args=(r,), daemon=True) for r in range(n)]
        for t in threads:
            t.start()
        try:
            for i in range(len(chunks)):
                item = queues[i % n].get()
                if isinstance(item, BaseException):
                    raise item
                yield item
            if mine and self.index_path and not self._index_saved:
                for t in threads:
                    t.join()
                for f in files:
                    f.save_index(self.index_path, merge_existing=True)
                self._index_saved = True
        finally:
            stop.set()
            while any(t.is_alive() for t in threads):
                for q in queues:
                    try:
                        q.get_nowait()
                    except queue.Empty:
                        pass
                threads[0].join(0.005)
            if mine:
                self._busy.release()
            else:
                for f in files:
                    f.close()


class DataH5womd:
    """`data_h5_womd.py:58-241` without Lightning: same arguments, tensor tables and loader names.  `num_workers` is accepted and
    unused (two native reader threads per loader, each fanning its decode out to `TB_H5_THREADS` workers, replace the worker processes)."""

    def __init__(self, data_dir: str, filename_train: str = "training", filename_val: str = "validation", filename_test: str = "testing",
                 batch_size: int = 3, num_workers: int = 4, n_agent: int = 64, n_pl: int = 1024, n_tl_stop: int = 40,
                 rank: int = 0, world_size: int = 1, index_dir: Optional[str] = None) -> None:
        self.interactive_challenge = "interactive" in filename_val or "interactive" in filename_test
        self.path_train_h5 = f"{data_dir}/{filename_train}.h5"
        self.path_val_h5 = f"{data_dir}/{filename_val}.h5"
        self.path_test_h5 = f"{data_dir}/{filename_test}.h5"
        self.batch_size, self.num_workers, self.rank, self.world_size = batch_size, num_workers, rank, world_size
        self.index_dir = index_dir  # where the loaders keep "<file>.r<rank>of<world>.tbidx" chunk indices (None: in memory only)
        n_step, n_hist, n_no_sim, n_tl, n_node = 91, 11, 256, 100, 20

        def agent_tables(pre: str, s: int, who: str, n: int, full: bool) -> Dict[str, Tuple[int, ...]]:
            t = {f"{pre}{who}/valid": (s, n), f"{pre}{who}/pos": (s, n, 2), f"{pre}{who}/z": (s, n, 1), f"{pre}{who}/vel": (s, n, 2),
                 f"{pre}{who}/spd": (s, n, 1), f"{pre}{who}/yaw_bbox": (s, n, 1), f"{pre}{who}/type": (n, 3), f"{pre}{who}/size": (n, 3)}
            if full:
                t.update({f"{pre}{who}/acc": (s, n, 1), f"{pre}{who}/yaw_rate": (s, n, 1), f"{pre}{who}/role": (n, 3)})
            return t

        def tl_tables(pre: str, s: int) -> Dict[str, Tuple[int, ...]]:
            return {f"{pre}tl_lane/valid": (s, n_tl), f"{pre}tl_lane/state": (s, n_tl, 5), f"{pre}tl_lane/idx": (s, n_tl),
                    f"{pre}tl_stop/valid": (s, n_tl_stop), f"{pre}tl_stop/state": (s, n_tl_stop, 5), f"{pre}tl_stop/pos": (s, n_tl_stop, 2),
                    f"{pre}tl_stop/dir": (s, n_tl_stop, 2)}

        map_tables = {"map/valid": (n_pl, n_node), "map/type": (n_pl, 11), "map/pos": (n_pl, n_node, 2), "map/dir": (n_pl, n_node, 2),
                      "map/boundary": (4,)}
        self.tensor_size_train = {**agent_tables("", n_step, "agent", n_agent, True), "agent/cmd": (n_agent, 8), "agent/goal": (n_agent, 4),
                                  "agent/dest": (n_agent,), **map_tables, **tl_tables("", n_step)}
        self.tensor_size_test = {"history/agent/object_id": (n_agent,), "history/agent_no_sim/object_id": (n_no_sim,),
                                 **agent_tables("history/", n_hist, "agent", n_agent, True),
                                 **agent_tables("history/", n_hist, "agent_no_sim", n_no_sim, False), **map_tables, **tl_tables("history/", n_hist)}
        self.tensor_size_val = {"agent/object_id": (n_agent,), "agent_no_sim/object_id": (n_no_sim,),
                                **agent_tables("", n_step, "agent_no_sim", n_no_sim, False), **self.tensor_size_train, **self.tensor_size_test}
        self.train_dataset = self.val_dataset = self.test_dataset = None

    def setup(self, stage: Optional[str] = None) -> None:
        def mk(path: str, size: Dict, split: str) -> PackedSceneLoader:
            idx = f"{self.index_dir}/{os.path.basename(path)}.r{self.rank}of{self.world_size}.tbidx" if self.index_dir else None
            return PackedSceneLoader(path, size, split, self.batch_size, rank=self.rank, world_size=self.world_size, index_path=idx)

        if stage == "fit" or stage is None:
            self.train_dataset = mk(self.path_train_h5, self.tensor_size_train, "train")
            self.val_dataset = mk(self.path_val_h5, self.tensor_size_val, "val")
        elif stage == "validate":
            self.val_dataset = mk(self.path_val_h5, self.tensor_size_val, "val")
        elif stage == "test":
            self.test_dataset = mk(self.path_test_h5, self.tensor_size_test, "test")

    def train_dataloader(self) -> PackedSceneLoader:
        return self.train_dataset

    def val_dataloader(self) -> PackedSceneLoader:
        return self.val_dataset

    def test_dataloader(self) -> PackedSceneLoader:
        return self.test_dataset


# ------------------------------------------------------------------------------------------------ writer
def write_packed_h5(path: str, episodes: Sequence[Dict[str, np.ndarray]], attrs: Optional[Sequence[Dict]] = None, deflate: int = 4,
                    shuffle: bool = True, chunk_div: int = 1) -> None:
    """Write per-episode tensor dicts (numpy: bool, float32, int64) in the format of `pack_h5_womd.py:378-392`; storage options as
    tb_h5_writer_options."""
    lib = load()
    w = C.c_void_p()
    _check(lib.tb_h5_writer_open(path.encode(), C.byref(w)), f"tb_h5_writer_open({path})")
    _check(lib.tb_h5_writer_options(w, deflate, int(shuffle), chunk_div), "tb_h5_writer_options")
    for i, ep in enumerate(episodes):
        a = attrs[i] if attrs is not None else {}
        center = np.asarray(a.get("scenario_center", [0.0, 0.0]), dtype=np.float64)
        _check(lib.tb_h5_writer_episode(w, i, str(a.get("scenario_id", f"synthetic_{i}")).encode(),
                                        center.ctypes.data_as(C.POINTER(C.c_double)), len(center), float(a.get("scenario_yaw", 0.0)),
                                        int(a.get("with_map", True))), "tb_h5_writer_episode")
        for k, v in ep.items():
            v = np.ascontiguousarray(v)
            if v.dtype == np.bool_:
                kind, v = MASK_U8, v.view(np.uint8)
            elif v.dtype == np.float32:
                kind = F32
            elif v.dtype == np.int64:
                kind = I64
            else:
                raise TypeError(f"{k}: dtype {v.dtype} is not one the packed format holds (bool, float32, int64)")
            dims = (C.c_int64 * max(v.ndim, 1))(*v.shape)
            _check(lib.tb_h5_writer_dataset(w, k.encode(), kind, dims, v.ndim, C.c_void_p(v.ctypes.data)), f"tb_h5_writer_dataset({k})")
    _check(lib.tb_h5_writer_close(w, len(episodes)), "tb_h5_writer_close")
