"""Import the reference (`/root/reference/src`) in THIS container for golden generation.

Only `tools/gen_golden.py` uses this; nothing under `tests/`, `bench.py` or the package imports
it, and it never runs on the GPU box (the reference does not exist there).  The reference needs
hydra / omegaconf / pytorch_lightning / torchmetrics / wandb / transforms3d / tensorflow /
waymo_open_dataset / cv2 / gym, none of which are installed: the few names it touches on the
inference path are stubbed in `sys.modules` (recipe: SURVEY.md Appendix C).  Nothing is written
under /root/reference (`sys.dont_write_bytecode`).
"""
from __future__ import annotations

import importlib
import inspect
import sys
import types

import torch
from torch import nn

REF_SRC = "/root/reference/src"


class AttrDict(dict):
    """dict with attribute access, standing in for omegaconf.DictConfig."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def to_attr(x):
    if isinstance(x, dict):
        return AttrDict({k: to_attr(v) for k, v in x.items()})
    if isinstance(x, (list, tuple)):
        return [to_attr(v) for v in x]
    return x


def _get_class(path: str):
    mod, _, name = path.rpartition(".")
    return getattr(importlib.import_module(mod), name)


def _instantiate(cfg, *args, **kwargs):
    cfg = dict(cfg)
    target = cfg.pop("_target_")
    for k in ("_recursive_", "_convert_", "_partial_"):
        cfg.pop(k, None)
        kwargs.pop(k, None)
    merged = {k: to_attr(v) for k, v in cfg.items()}
    merged.update(kwargs)
    return _get_class(target)(*args, **merged)


class _LightningModule(nn.Module):
    def __init__(self, *a, **k):
        super().__init__()
        self.current_epoch = 0
        self.global_rank = 0
        self.logger = None

    def save_hyperparameters(self):
        frame = inspect.currentframe().f_back
        sig = inspect.signature(type(self).__init__)
        hp = AttrDict()
        for name in sig.parameters:
            if name == "self":
                continue
            hp[name] = to_attr(frame.f_locals[name])
        object.__setattr__(self, "_hparams", hp)

    @property
    def hparams(self):
        return self._hparams

    def log(self, *a, **k):
        pass


class _Metric(nn.Module):
    def __init__(self, *a, **k):
        super().__init__()

    def add_state(self, name, default, dist_reduce_fx=None):
        self.__dict__.setdefault("_tb_defaults", {})[name] = default.clone() if torch.is_tensor(default) else default
        setattr(self, name, default)

    # torchmetrics' `Metric.forward` on a freshly reset metric = update on zero states + compute, and `reset` = back to the
    # defaults: what `training_step` uses (`waymo_motion.py:403-417`; tools/train_reference.py)
    def reset(self):
        for name, default in self.__dict__.get("_tb_defaults", {}).items():
            setattr(self, name, default.clone() if torch.is_tensor(default) else default)

    def forward(self, *a, **k):
        self.reset()
        self.update(*a, **k)
        return self.compute()


class _Dummy:
    def __init__(self, *a, **k):
        pass


class _DummyModule(nn.Module):
    def __init__(self, *a, **k):
        super().__init__()


def install() -> None:
    """Install the stubs and put the reference sources on sys.path (idempotent)."""
    sys.dont_write_bytecode = True
    if "pl_modules.waymo_motion" in sys.modules:
        return

    def mod(name, **attrs):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    mod("omegaconf", DictConfig=AttrDict, ListConfig=list)
    hu = mod("hydra.utils", instantiate=_instantiate, get_class=_get_class)
    mod("hydra", utils=hu)
    mod("transforms3d.euler")
    mod("transforms3d", euler=sys.modules["transforms3d.euler"])
    mod("wandb")
    pl_loggers = mod("pytorch_lightning.loggers", WandbLogger=_Dummy)
    mod("pytorch_lightning", LightningModule=_LightningModule, loggers=pl_loggers)
    tm_metric = mod("torchmetrics.metric", Metric=_Metric)
    mod("torchmetrics", metric=tm_metric, Metric=_Metric)

    if REF_SRC not in sys.path:
        sys.path.insert(0, REF_SRC)

    # heavy leaf modules of the reference that need TF / waymo protos / cv2 / gym: replace
    # them wholesale (they sit after the hot path)
    importlib.import_module("models.metrics")
    importlib.import_module("utils")
    mod("models.metrics.womd", WOMDMetrics=_DummyModule)
    mod("utils.submission", SubWOMD=_Dummy)
    mod("utils.vis_waymo", VisWaymo=_Dummy)
    importlib.import_module("pl_modules.waymo_motion")


def build_reference(cfg: dict, n_agent: int, n_pl: int, n_tl: int = 40) -> nn.Module:
    """Construct the reference `WaymoMotion` (`src/pl_modules/waymo_motion.py:27`) from a resolved
    plain-dict config (`trafficbots_amd.config`) in eval mode."""
    install()
    from pl_modules.waymo_motion import WaymoMotion

    n_step, n_step_hist, n_node = 91, 11, 20
    data_size = {
        "agent/valid": (n_step, n_agent),
        "agent/pos": (n_step, n_agent, 2),
        "agent/vel": (n_step, n_agent, 2),
        "agent/spd": (n_step, n_agent, 1),
        "agent/acc": (n_step, n_agent, 1),
        "agent/yaw_bbox": (n_step, n_agent, 1),
        "agent/yaw_rate": (n_step, n_agent, 1),
        "agent/type": (n_agent, 3),
        "agent/size": (n_agent, 3),
        "map/valid": (n_pl, n_node),
        "map/type": (n_pl, 11),
        "map/pos": (n_pl, n_node, 2),
        "map/dir": (n_pl, n_node, 2),
        "tl_stop/valid": (n_step, n_tl),
        "tl_stop/state": (n_step, n_tl, 5),
        "tl_stop/pos": (n_step, n_tl, 2),
        "tl_stop/dir": (n_step, n_tl, 2),
    }
    c = to_attr(cfg)
    # groups the reference constructor wants but the hot path never reads
    full = dict(
        time_step_current=c.time_step_current,
        time_step_gt=c.time_step_gt,
        time_step_end=c.time_step_end,
        time_step_sim_start=c.time_step_sim_start,
        hidden_dim=c.hidden_dim,
        data_size=to_attr(data_size),
        pre_processing=AttrDict(
            scene_centric=AttrDict(_target_="data_modules.scene_centric.SceneCentricPreProcessing"),
            input=AttrDict(_target_="data_modules.sc_input.SceneCentricInput", **c.pre_processing.input),
            latent=AttrDict(
                _target_="data_modules.sc_latent.SceneCentricLatent",
                pe_dim=c.pre_processing.input.pe_dim,
                pose_pe=c.pre_processing.input.pose_pe,
                perturb_input_to_latent=bool(cfg.get("pre_processing", {}).get("latent", {}).get("perturb_input_to_latent", False)),
                dropout_p_history=cfg.get("pre_processing", {}).get("latent", {}).get("dropout_p_history", -1),
                max_meter=50.0,
                max_rad=3.14,
            ),
        ),
        step_detach_hidden=-1,
        model=AttrDict(
            _target_="models.traffic_bots.TrafficBots",
            final_mlp=AttrDict(use_layernorm=False, activation="relu", dropout_p=0.1),
            **c.model,
        ),
        p_training_rollout_prior=0.1,
        detach_state_policy=c.detach_state_policy,
        training_deterministic_action=True,
        differentiable_reward=c.differentiable_reward,
        p_drop_hidden=cfg.get("p_drop_hidden", -1.0),
        n_video_batch=0,
        n_joint_future=c.n_joint_future,
        waymo_post_processing=AttrDict(
            k_pred=6, use_ade=True, score_temperature=1e2, mpa_nms_thresh=[], mtr_nms_thresh=[], aggr_thresh=[], n_iter_em=3
        ),
        dynamics=c.dynamics,
        action_head=c.action_head,
        teacher_forcing_training=AttrDict(step_spawn_agent=10, step_warm_start=10),
        teacher_forcing_reactive_replay=c.teacher_forcing_reactive_replay,
        teacher_forcing_joint_future_pred=c.teacher_forcing_joint_future_pred,
        training_metrics=c.training_metrics,
        traffic_rule_checker=c.traffic_rule_checker,
        optimizer=AttrDict(),
        lr_scheduler=None,
        lr_goal=3e-4,
        sub_womd_reactive_replay=AttrDict(),
        sub_womd_joint_future_pred=AttrDict(),
    )
    model = WaymoMotion(**full)
    model.eval()
    return model


# ---------------------------------------------------------------------------------------------------------------------
# h5py stand-in (the image has the HDF5 C library but no h5py): just enough of h5py's object model for the reference's
# dataset classes (`src/data_modules/data_h5_womd.py:9-55`) -- File as a context manager, `hf.attrs[...]`, `hf[key]` groups,
# `group.attrs[...]`, `group[key]` datasets that `np.ascontiguousarray` can take.  Values come from a reader object handed in
# by the caller (`reader_cls(filepath)` with `.dataset(path) -> ndarray`, `.attr(obj, name)`, `.close()`): tests/golden/
# gen_h5_reference.py passes the ctypes -> libhdf5 reader, so that the reference's OWN `__getitem__` logic runs on real files.
def install_h5py(reader_cls) -> None:
    import numpy as np

    class _Attrs:
        def __init__(self, reader, obj):
            self._r, self._obj = reader, obj

        def __getitem__(self, name):
            return self._r.attr(self._obj, name)

    class _Dataset:
        def __init__(self, reader, path):
            self._r, self._path = reader, path

        def __array__(self, dtype=None, copy=None):
            a = self._r.dataset(self._path)
            return a if dtype is None else a.astype(dtype)

        @property
        def shape(self):
            return np.asarray(self).shape

    class _Group:
        def __init__(self, reader, path):
            self._r, self._path = reader, path
            self.attrs = _Attrs(reader, path or "/")

        def __getitem__(self, key):
            path = f"{self._path}/{key}" if self._path else str(key)
            try:
                self._r.dataset(path)  # a dataset if it can be opened as one, a group otherwise
            except (KeyError, OSError, TypeError):
                return _Group(self._r, path)
            return _Dataset(self._r, path)

    class File(_Group):
        def __init__(self, filepath, mode="r", libver=None, swmr=False):
            assert mode == "r"
            super().__init__(reader_cls(filepath), "")

        def __enter__(self):
            return self

        def __exit__(self, *exc):
            self._r.close()
            return False

    m = types.ModuleType("h5py")
    m.File = File
    sys.modules["h5py"] = m


def import_reference_datasets(reader_cls):
    """`data_modules.data_h5_womd` of the reference with the h5py stand-in in place (build container only)."""
    install()
    install_h5py(reader_cls)
    pl = sys.modules["pytorch_lightning"]
    if not hasattr(pl, "LightningDataModule"):
        pl.LightningDataModule = type("LightningDataModule", (), {"__init__": lambda self, *a, **k: None})
    return importlib.import_module("data_modules.data_h5_womd")
