"""Host side of the HIP hot path: owns a `tb_ctx`, feeds it borrowed PyTorch-ROCm buffers.

PyTorch is plumbing here (device memory, the current HIP stream); all arithmetic of the path
runs in `libtrafficbots_hip.so`.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Tuple

import numpy as np
import torch
from torch import Tensor

from . import hip

N_PL_NODE = 20


def as_bool(x: Tensor) -> Tensor:
    """uint8 0/1 mask (what the kernels write and the staging makes) -> bool WITHOUT a conversion kernel: a reinterpreting view."""
    return x.view(torch.bool) if x.dtype == torch.uint8 else x.bool()


def as_u8(x: Tensor) -> Tensor:
    """bool -> uint8 as a view (contiguous input), else a conversion."""
    if x.dtype == torch.uint8:
        return x.contiguous()
    if x.dtype == torch.bool and x.is_contiguous():
        return x.view(torch.uint8)
    return x.to(torch.uint8).contiguous()


def _onehot_to_index(x: Tensor) -> Tensor:
    """bool one-hot [..., C] -> int32 index, -1 where no class is set."""
    idx = x.to(torch.int32).argmax(-1).to(torch.int32)
    return torch.where(x.any(-1), idx, torch.full_like(idx, -1)).contiguous()


def teacher_forcing_mask(valid: Tensor, step_spawn_agent: int = 10, step_warm_start: int = 10) -> Tensor:
    """Mask of `TeacherForcing.get` (`src/utils/teacher_forcing.py:33-74`) with the schedule terms
    (step_horizon, prob_forcing_agent) at their default 0: always spawn at step 0, spawn on
    invalid->valid transitions up to `step_spawn_agent`, force every valid agent up to `step_warm_start`.
    valid: [B, S, A] bool."""
    m = torch.zeros_like(valid)
    m[:, 0] |= valid[:, 0]
    if step_spawn_agent > 0:
        sp = (~valid[:, :-1]) & valid[:, 1:]
        sp[:, step_spawn_agent:] = False
        m[:, 1:] |= sp
    if step_warm_start >= 0:
        m[:, : step_warm_start + 1] |= valid[:, : step_warm_start + 1]
    # provenance for `warm_start_promise_holds`: the warm-start length AND the tensor's version counter at this point -- an in-place
    # edit afterwards (`mask[:, 5, i] = False`, the reference's own `mask_state_override[:, 0] = True` idiom on a view) bumps the
    # counter and voids the tag, so an edited mask is checked on its data like any caller-made one
    m._tb_warm_start = (int(step_warm_start), m._version)
    return m


def warm_start_promise_holds(mask_teacher_forcing: Tensor, valid: Tensor, w: int) -> bool:
    """`tb_rollout_io.warm_start_steps` = w promises that every valid agent is teacher-forced at steps 0..w.  True without a look at
    the data when the mask came out of :func:`teacher_forcing_mask` with step_warm_start >= w; a caller-made mask is checked
    (one device -> host read)."""
    if w <= 0:
        return True
    tag = getattr(mask_teacher_forcing, "_tb_warm_start", None)
    if tag is not None and tag[0] >= w and tag[1] == mask_teacher_forcing._version:
        return True
    m, v = mask_teacher_forcing[:, : w + 1].bool(), valid[:, : w + 1].bool().to(mask_teacher_forcing.device)
    return bool((m | ~v).all())


METRIC_FIELDS = ("err_counter", "err_pos_meter", "err_rot_deg", "err_spd_m_per_s", "counter_agent", "counter_veh", "outside_map",
                 "collided", "run_road_edge", "run_red_light", "passive", "goal_reached", "dest_reached")
RULE_KEYS = ("collided", "collided_this_step", "run_road_edge", "run_road_edge_this_step", "run_red_light",
             "run_red_light_this_step", "passive", "passive_this_step")


def no_early_exit(valid, n_steps: int) -> bool:
    """True when no agent turns from valid to invalid within the first `n_steps` steps of `valid` [B,S,A] (evaluated where the
    batch still lives, normally on the host before the upload): the condition under which the warm start can be batched
    (`tb_rollout_io.warm_start_steps`)."""
    v = torch.as_tensor(valid)[:, :n_steps].bool()
    return not bool((v[:, :-1] & ~v[:, 1:]).any())


_STAGERS: Dict = {}


def default_stager(device, n_hist: int = 11, tf_params: Tuple[int, int] = (10, 10)):
    """The process-wide `staging.HostStager` of (device, n_hist, teacher-forcing parameters): pinned slabs are worth keeping."""
    from .staging import HostStager

    key = (str(torch.device(device)), int(n_hist), tuple(int(x) for x in tf_params))
    if key not in _STAGERS:
        _STAGERS[key] = HostStager(device, n_hist, tf_params)
    return _STAGERS[key]


def scene_from_batch(batch: Dict[str, Tensor], device, n_hist: int = 11, with_gt: bool = False) -> Dict[str, Tensor]:
    """Reference batch (`data_h5_womd.py:85-157`, bool tensors) -> the C ABI's device layout (uint8 masks, int32 class
    indices, yaw/spd/acc as [B,NH,A]).  Test / validation batches carry the history under "history/*"; a TRAINING-split batch
    has only "agent/*" / "tl_stop/*" over 91 steps, of which `SceneCentricPreProcessing` (prefix "" in training,
    `scene_centric.py:92-121`) takes the first `n_hist` steps -- same here.  `warm_ok` (a Python bool, not a tensor) records
    that no agent leaves within the history, see :func:`no_early_exit`.

    A batch that lives on the HOST (numpy arrays / CPU tensors -- what a DataLoader yields) takes the staged path (round 6,
    `staging.HostStager`): the layout conversion happens on the host while the batch is copied into ONE pinned slab, one
    host-to-device copy, every returned tensor a view of one device buffer, no device-side conversion kernels; `with_gt` keeps the
    nested `scene["gt"]` of a validation / training batch (what :func:`gt_from_batch` builds).  A batch that is already on the
    device is converted there with torch ops (:func:`scene_from_batch_torch`)."""
    from .staging import is_host_batch

    if is_host_batch(batch):
        scene = default_stager(device, n_hist).stage(batch)
        if not with_gt:
            scene.pop("gt", None)
        return scene
    return scene_from_batch_torch(batch, device, n_hist)


def scene_from_batch_torch(batch: Dict[str, Tensor], device, n_hist: int = 11) -> Dict[str, Tensor]:
    """:func:`scene_from_batch` with torch ops on `device` (one copy per tensor + conversion kernels): the path of a batch that is
    already device-resident, and the cross-check of the staged path in the CPU tests."""
    pre = "history/" if "history/agent/valid" in batch else ""
    if not pre and "agent/valid" not in batch:
        raise KeyError("batch carries neither 'history/agent/*' nor 'agent/*'")

    def g(k):
        v = batch[k]
        if isinstance(v, np.ndarray):
            v = torch.from_numpy(v)
        return v.to(device)

    def h(k):  # history part of a per-step tensor
        return g(pre + k)[:, :n_hist]

    f32, u8 = torch.float32, torch.uint8
    s: Dict[str, Tensor] = {}
    s["warm_ok"] = no_early_exit(batch[pre + "agent/valid"], n_hist)
    s["agent_valid"] = h("agent/valid").to(u8).contiguous()
    pos = h("agent/pos").to(f32)
    yaw = h("agent/yaw_bbox").to(f32)
    spd = h("agent/spd").to(f32)
    s["agent_pos"] = pos.contiguous()
    s["agent_yaw"] = yaw[..., 0].contiguous()
    s["agent_spd"] = spd[..., 0].contiguous()
    s["agent_state"] = torch.cat([pos, yaw, spd], -1).contiguous()
    s["agent_vel"] = h("agent/vel").to(f32).contiguous()
    s["agent_acc"] = h("agent/acc")[..., 0].to(f32).contiguous()
    s["agent_yaw_rate"] = h("agent/yaw_rate")[..., 0].to(f32).contiguous()
    s["agent_type"] = _onehot_to_index(g(pre + "agent/type"))
    s["agent_size"] = g(pre + "agent/size").to(f32).contiguous()
    s["map_valid"] = g("map/valid").to(u8).contiguous()
    s["map_type"] = _onehot_to_index(g("map/type"))
    s["map_pos"] = g("map/pos").to(f32).contiguous()
    s["map_dir"] = g("map/dir").to(f32).contiguous()
    s["map_boundary"] = g("map/boundary").to(f32).contiguous()
    s["tl_valid"] = h("tl_stop/valid").to(u8).contiguous()
    s["tl_state"] = _onehot_to_index(h("tl_stop/state"))
    s["tl_pos"] = h("tl_stop/pos").to(f32).contiguous()
    s["tl_dir"] = h("tl_stop/dir").to(f32).contiguous()
    return s


TRAIN_FIELDS = ("vae_kl_counter", "vae_kl", "diffbar_reward_counter", "diffbar_reward", "goal_loss", "goal_counter")
_CRITERIA = {"SmoothL1Loss": 0, "MSELoss": 1, "L1Loss": 2}
_ANGULAR = {None: 0, "cast": 1, "cosine": 2, "vector": 3}


def history_len(batch: Dict) -> int:
    """Number of history steps a test-split / validation-split batch carries ("history/agent/valid" is [B, steps, A]); 0 without one."""
    v = batch.get("history/agent/valid")
    return int(v.shape[1]) if v is not None else 0


def hist_from_batch(batch: Dict[str, Tensor], device, n_hist: int, tf_params: Tuple[int, int]) -> Dict[str, Tensor]:
    """What `test_step` hands the ROLLOUT as `features["agent_valid" / "agent_state" / "vel" / "acc" / "yaw_rate"]`: the batch's
    history over ALL of its steps (`waymo_motion.py:925-926`: batch["agent/*"] = batch["history/agent/*"]; `:538-545`) -- while the
    encoders see the first time_step_current + 1 steps only (`scene_centric.py:92-121`).  With the default time_step_current = 10 the
    two coincide and this is not built; with a shorter one the teacher-forcing mask, the state overrides and the ground-truth validity
    the kill rule spares (`dynamics.py:161-167`) still reach to the end of the 11-step history (found by tools/
    fuzz_oracle_vs_reference.py, round 6).  Same layout as the agent part of :func:`gt_from_batch`; plain torch copies (a config off
    the default: not the staged fast path)."""

    def g(k):
        v = batch["history/agent/" + k]
        if isinstance(v, np.ndarray):
            v = torch.from_numpy(v)
        return v.to(device)

    f32, u8 = torch.float32, torch.uint8
    pos, yaw, spd = g("pos").to(f32), g("yaw_bbox").to(f32), g("spd").to(f32)
    valid = g("valid")
    s: Dict[str, Tensor] = {
        "warm_ok": no_early_exit(batch["history/agent/valid"], n_hist),
        "agent_valid": valid.to(u8).contiguous(), "agent_state": torch.cat([pos, yaw, spd], -1).contiguous(),
        "agent_vel": g("vel").to(f32).contiguous(), "agent_acc": g("acc")[..., 0].to(f32).contiguous(),
        "agent_yaw_rate": g("yaw_rate")[..., 0].to(f32).contiguous(),
    }
    s["_tf_mask"] = as_u8(teacher_forcing_mask(valid.bool(), *tf_params))
    s["_tf_params"] = tuple(tf_params)
    return s


def gt_from_batch(batch: Dict[str, Tensor], device, n_hist: int = 11) -> Dict[str, Tensor]:
    """Ground-truth part of a validation / training batch (`data_h5_womd.py:85-118`: "agent/*", "tl_stop/*" over all 91
    steps) -> the C ABI's device layout: what `SceneCentricPreProcessing` exposes as "gt/*" (`scene_centric.py:103-110`),
    `SceneCentricLatent` as "latent_post/*" (`sc_latent.py:150-163,196-217`) and `reactive_replay` as `features`
    (`waymo_motion.py:448-462`)."""

    def g(k):
        v = batch[k]
        if isinstance(v, np.ndarray):
            v = torch.from_numpy(v)
        return v.to(device)

    f32, u8 = torch.float32, torch.uint8
    s: Dict[str, Tensor] = {}
    s["warm_ok"] = no_early_exit(batch["agent/valid"], n_hist)  # (time_step_current + 1 steps: the warm start of every TeacherForcing config)
    s["agent_valid"] = g("agent/valid").to(u8).contiguous()
    pos, yaw, spd = g("agent/pos").to(f32), g("agent/yaw_bbox").to(f32), g("agent/spd").to(f32)
    s["agent_pos"] = pos.contiguous()
    s["agent_yaw"] = yaw[..., 0].contiguous()
    s["agent_spd"] = spd[..., 0].contiguous()
    s["agent_state"] = torch.cat([pos, yaw, spd], -1).contiguous()
    s["agent_vel"] = g("agent/vel").to(f32).contiguous()
    s["agent_acc"] = g("agent/acc")[..., 0].to(f32).contiguous()
    s["agent_yaw_rate"] = g("agent/yaw_rate")[..., 0].to(f32).contiguous()
    s["agent_type"] = _onehot_to_index(g("agent/type"))
    s["agent_size"] = g("agent/size").to(f32).contiguous()
    s["agent_role"] = g("agent/role").to(u8).contiguous()
    s["tl_valid"] = g("tl_stop/valid").to(u8).contiguous()
    s["tl_state"] = _onehot_to_index(g("tl_stop/state"))
    s["tl_pos"] = g("tl_stop/pos").to(f32).contiguous()
    s["tl_dir"] = g("tl_stop/dir").to(f32).contiguous()
    s["gt_dest"] = g("agent/dest").to(torch.int32).contiguous()
    if "agent/goal" in batch:
        s["gt_goal"] = g("agent/goal").to(f32).contiguous()
    return s


class HipEngine:
    """One `tb_ctx` on one device."""

    def __init__(self, cfg: Dict, device: str = "cuda:0") -> None:
        if not torch.cuda.is_available():
            raise RuntimeError("trafficbots_amd: no HIP device visible to PyTorch-ROCm (no CPU fallback)")
        self.lib = hip.load()
        self.cfg = cfg
        self.device = torch.device(device)
        torch.cuda.set_device(self.device)
        self._ctx = C.c_void_p()
        c = hip.make_config(cfg)
        rc = self.lib.tb_create(C.byref(c), C.byref(self._ctx))
        if rc != 0:
            raise RuntimeError(f"tb_create failed ({rc})")
        self.n_hist = cfg["time_step_current"] + 1
        self._weights_loaded = False

    def __del__(self):
        try:
            if getattr(self, "_ctx", None) and self._ctx.value:
                self.lib.tb_destroy(self._ctx)
                self._ctx = C.c_void_p()
        except Exception:
            pass

    def _check(self, rc: int, what: str) -> None:
        if hip.guard_hook is not None:  # (test hook: copy the guarded shadow buffers of this call back into the caller's tensors)
            hip.guard_hook.writeback()
        if rc != 0:
            raise RuntimeError(f"{what} failed: {self.lib.tb_last_error(self._ctx).decode()}")

    def _stream(self) -> C.c_void_p:
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def stager(self, tf_params: Tuple[int, int] = (10, 10)):
        """The pinned-slab stager of this engine's device (`staging.HostStager`; shared per (device, n_hist, teacher-forcing
        parameters) in the process)."""
        return default_stager(self.device, self.n_hist, tf_params)

    # -------------------------------------------------------------------------------- weights
    def load_state_dict(self, state_dict: Dict) -> None:
        """Reference-format `state_dict` (numpy or torch values, SURVEY Appendix B names)."""
        for k, v in state_dict.items():
            if isinstance(v, Tensor):
                v = v.detach().cpu().numpy()
            v = np.ascontiguousarray(v, dtype=np.float32)
            self._check(
                self.lib.tb_load_weight(self._ctx, k.encode(), v.ctypes.data_as(hip.c_f32p), v.size), f"tb_load_weight({k})"
            )
        self._check(self.lib.tb_finalize_weights(self._ctx, self._stream()), "tb_finalize_weights")
        self._weights_loaded = True

    # -------------------------------------------------------------------------------- encoders
    def encode_scene(self, s: Dict[str, Tensor]) -> Dict[str, Tensor]:
        b, nh, a = s["agent_valid"].shape
        p = s["map_valid"].shape[1]
        t = s["tl_valid"].shape[2]
        dev, f32, u8 = self.device, torch.float32, torch.uint8
        out = {
            "map_feature": torch.empty(b, p, 128, device=dev, dtype=f32),
            "map_feature_valid": torch.empty(b, p, device=dev, dtype=u8),
            "agent_feature": torch.empty(b, nh, a, 128, device=dev, dtype=f32),
            "tl_feature": torch.empty(b, nh, t, 128, device=dev, dtype=f32),
            "latent_mean": torch.empty(b, a, 16, device=dev, dtype=f32),
            "latent_valid": torch.empty(b, a, device=dev, dtype=u8),
            "dest_logits": torch.empty(b, a, p, device=dev, dtype=f32),
        }
        io = hip.TbEncodeIO()
        io.n_scene, io.n_agent, io.n_pl, io.n_tl, io.n_hist = b, a, p, t, nh
        for name in ("agent_valid", "map_valid", "tl_valid"):
            setattr(io, name, hip.ptr(s[name], hip.c_u8p))
        # (a scene made from the reference's own attr / pe tensors carries "ext_*" entries and no raw fields for that token kind)
        for name in ("agent_pos", "agent_yaw", "agent_vel", "agent_spd", "agent_acc", "agent_yaw_rate", "agent_size",
                     "map_pos", "map_dir", "tl_pos", "tl_dir"):
            setattr(io, name, hip.ptr(s.get(name), hip.c_f32p))
        for name in ("agent_type", "map_type", "tl_state"):
            setattr(io, name, hip.ptr(s.get(name), hip.c_i32p))
        for name in ("ext_agent_attr", "ext_agent_pe", "ext_map_attr", "ext_map_pe", "ext_tl_attr", "ext_tl_pe"):
            setattr(io, name, hip.ptr(s.get(name), hip.c_f32p))
        out["_keepalive"] = s
        for name in ("map_feature", "agent_feature", "tl_feature", "latent_mean", "dest_logits"):
            setattr(io, name, hip.ptr(out[name], hip.c_f32p))
        io.map_feature_valid = hip.ptr(out["map_feature_valid"], hip.c_u8p)
        io.latent_valid = hip.ptr(out["latent_valid"], hip.c_u8p)
        self._check(self.lib.tb_encode_scene(self._ctx, C.byref(io), self._stream()), "tb_encode_scene")
        return out

    def encode_posterior(self, gt: Dict[str, Tensor], feats: Dict[str, Tensor]) -> Dict[str, Tensor]:
        """Posterior personality over the full ground-truth episode (`tb_encode_posterior`; `LatentEncoder.forward(posterior=
        True)`, `latent_encoder.py:98-136`, on the "latent_post/*" inputs): `gt` from :func:`gt_from_batch`, `feats` the output
        of :meth:`encode_scene` for the same scenes (its map feature is reused).  Returns latent_mean [B,A,16], latent_valid."""
        b, ns, a = gt["agent_valid"].shape
        p = feats["map_feature"].shape[1]
        t = gt["tl_valid"].shape[2]
        out = {"latent_mean": torch.empty(b, a, 16, device=self.device, dtype=torch.float32),
               "latent_valid": torch.empty(b, a, device=self.device, dtype=torch.uint8)}
        io = hip.TbPosteriorIO()
        io.n_scene, io.n_agent, io.n_pl, io.n_tl, io.n_step = b, a, p, t, ns
        for name in ("agent_valid", "tl_valid"):
            setattr(io, name, hip.ptr(gt[name], hip.c_u8p))
        for name in ("agent_pos", "agent_yaw", "agent_vel", "agent_spd", "agent_acc", "agent_yaw_rate", "agent_size", "tl_pos", "tl_dir"):
            setattr(io, name, hip.ptr(gt[name], hip.c_f32p))
        for name in ("agent_type", "tl_state"):
            setattr(io, name, hip.ptr(gt[name], hip.c_i32p))
        io.map_feature = hip.ptr(feats["map_feature"], hip.c_f32p)
        io.map_feature_valid = hip.ptr(feats["map_feature_valid"], hip.c_u8p)
        io.latent_mean = hip.ptr(out["latent_mean"], hip.c_f32p)
        io.latent_valid = hip.ptr(out["latent_valid"], hip.c_u8p)
        out["_keepalive"] = (gt, feats)
        self._check(self.lib.tb_encode_posterior(self._ctx, C.byref(io), self._stream()), "tb_encode_posterior")
        return out

    def train_partials(self, buf: Dict[str, Tensor], gt_valid: Optional[Tensor], gt_states: Optional[Tensor], agent_size: Tensor,
                       dest_logits: Optional[Tensor] = None, goal_valid: Optional[Tensor] = None, gt_dest: Optional[Tensor] = None,
                       post: Optional[Dict[str, Tensor]] = None, prior: Optional[Dict[str, Tensor]] = None,
                       agent_role: Optional[Tensor] = None, irrelevant_draw: Optional[Tensor] = None, generator=None):
        """`training_metrics.p_loss_for_irrelevant` = p > 0 (`training.py:85-89`): agents without a role lose their loss terms and a
        Bernoulli(p) draw per agent gets them back for ALL steps; the draw is `irrelevant_draw` [B,A] (explicit, as the golden passes it)
        or `torch.bernoulli(generator=generator)` here; `agent_role` [B,A,3] is required then.
        Forward training losses of a replayed episode (`tb_train_partials`): the per-step `DifferentiableReward`
        (`rewards.py:33-131`, config group `differentiable_reward`) over the buffer `buf` (preds [B,A,S,4], valid,
        override_masks [B,A,S]; K = 1) against ground truth [B,A,S(,4)], and the six `TrainingMetrics` "sum" states
        (`training.py:62-139`, group `training_metrics`) as a float64 device vector in `TRAIN_FIELDS` order.
        Without destination logits / personalities only the rewards (and the two reward states) are produced.
        Returns (diffbar_rewards [B,A,S], diffbar_rewards_valid [B,A,S] uint8, states [6])."""
        rw, tm = self.cfg["differentiable_reward"], self.cfg["training_metrics"]
        if tm["w_relevant_agent"] > 0:
            raise NotImplementedError(
                "training_metrics.w_relevant_agent > 0 is rejected: the reference builds w_mask_rel [n_batch, n_agent] "
                "(src/models/metrics/training.py:95-97) and multiplies the reward tensor [n_batch, n_agent, n_step] by "
                "w_mask_rel.unsqueeze(1) = [n_batch, 1, n_agent] (training.py:124), which only broadcasts when n_step == n_agent and then "
                "weights step s by agent s's relevance -- there is no defined behaviour to reproduce (the default config has 0, "
                "configs/model/traffic_bots.yaml:216)")
        dev, u8, f32 = self.device, torch.uint8, torch.float32
        b, a, n_step = buf["valid"].shape
        io = hip.TbTrainIO()
        io.n_scene, io.n_agent, io.n_step, io.n_pl = b, a, n_step, (dest_logits.shape[-1] if dest_logits is not None else 1)
        io.w_collision = float(rw["w_collision"])
        io.reduce_collision_with_max = int(bool(rw["reduce_collsion_with_max"]))
        io.use_il_loss = int(bool(rw["use_il_loss"]))
        io.crit_pos, io.crit_rot, io.crit_spd = (_CRITERIA[rw[k]["criterion"]] for k in ("l_pos", "l_rot", "l_spd"))
        io.angular_type = _ANGULAR[rw["l_rot"].get("angular_type")]
        io.w_pos, io.w_rot, io.w_spd = (float(rw[k]["weight"]) for k in ("l_pos", "l_rot", "l_spd"))
        io.use_vae_kl = int(tm["w_vae_kl"] > 0 and post is not None and prior is not None)
        io.use_diffbar_reward = int(tm["w_diffbar_reward"] > 0)
        io.use_goal = int(tm["w_goal"] > 0 and dest_logits is not None)
        io.kl_for_unseen_agent = int(bool(tm["kl_for_unseen_agent"]))
        io.loss_for_teacher_forcing = int(bool(tm["loss_for_teacher_forcing"]))
        io.step_training_start = int(tm["step_training_start"])
        io.kl_balance_scale, io.kl_free_nats = float(tm["kl_balance_scale"]), float(tm["kl_free_nats"])
        keep = []

        def c8(t):
            t = t.to(dev).to(u8).contiguous()
            keep.append(t)
            return hip.ptr(t, hip.c_u8p)

        def cf(t):
            t = t.to(dev).to(f32).contiguous()
            keep.append(t)
            return hip.ptr(t, hip.c_f32p)

        io.pred_valid, io.pred_states, io.override_masks = c8(buf["valid"]), cf(buf["preds"]), c8(buf["override_masks"])
        io.gt_valid = c8(gt_valid) if gt_valid is not None else hip.ptr(None, hip.c_u8p)
        io.gt_states = cf(gt_states) if gt_states is not None else hip.ptr(None, hip.c_f32p)
        io.agent_size = cf(agent_size)
        if io.use_goal:
            io.dest_logits, io.goal_valid = cf(dest_logits), c8(goal_valid)
            gd = gt_dest.to(dev).to(torch.int32).contiguous()
            keep.append(gd)
            io.gt_dest = hip.ptr(gd, hip.c_i32p)
        # p_loss_for_irrelevant masks the reward states as well (training.py:85-115): whenever the role / draw are supplied they are
        # applied.  The reward-only call that rollout() makes (no role, no draw, no KL / goal terms) cannot apply them: it returns
        # None for the states instead of unmasked numbers; the rewards themselves are per step and unaffected
        states_valid = True
        if tm["p_loss_for_irrelevant"] > 0 and not (io.use_vae_kl or io.use_goal) and agent_role is None and irrelevant_draw is None:
            states_valid = False
        elif tm["p_loss_for_irrelevant"] > 0:
            if agent_role is None:
                raise ValueError("p_loss_for_irrelevant > 0 needs agent_role")
            if irrelevant_draw is None:
                irrelevant_draw = torch.bernoulli(torch.full((b, a), float(tm["p_loss_for_irrelevant"]), device=dev), generator=generator)
            io.relevant, io.irrelevant_draw = c8(agent_role.bool().any(-1)), c8(irrelevant_draw.reshape(b, a).bool())
        if io.use_vae_kl:
            io.post_mean, io.post_valid = cf(post["latent_mean"]), c8(post["latent_valid"])
            io.prior_mean, io.prior_valid = cf(prior["latent_mean"]), c8(prior["latent_valid"])
        rewards = torch.empty(b, a, n_step, device=dev, dtype=f32)
        rvalid = torch.empty(b, a, n_step, device=dev, dtype=u8)
        states = torch.zeros(len(TRAIN_FIELDS), device=dev, dtype=torch.float64)
        io.diffbar_rewards, io.diffbar_rewards_valid = hip.ptr(rewards, hip.c_f32p), hip.ptr(rvalid, hip.c_u8p)
        io.out = C.cast(C.c_void_p(states.data_ptr()), C.POINTER(C.c_double))
        self._check(self.lib.tb_train_partials(self._ctx, C.byref(io), self._stream()), "tb_train_partials")
        self._train_keepalive = keep
        return rewards, rvalid, (states if states_valid else None)

    # -------------------------------------------------------------------------------- rollout
    def rollout(
        self,
        s: Dict[str, Tensor],
        feats: Dict[str, Tensor],
        latent_sample: Optional[Tensor],
        latent_mean: Tensor,
        dest: Tensor,
        goal_valid: Tensor,
        k_futures: int,
        step_end: int,
        mask_teacher_forcing: Optional[Tensor] = None,
        tap_step: int = -1,
        out: Optional[Dict[str, Tensor]] = None,
        stepwise: bool = False,
        record_check_states: bool = False,
        gt: Optional[Dict[str, Tensor]] = None,
        latent_posterior: bool = False,
        warm_start_steps: int = 0,
        action_eps: Optional[Tensor] = None,
        hidden_drop=None,
        latent_eps: Optional[Tensor] = None,
        latent_deterministic: Optional[Tensor] = None,
        record_actions: bool = False,
    ) -> Dict[str, Tensor]:
        """`latent_sample=None`: the personalities are drawn by the rollout prologue itself (`MyDist.sample`, `distributions.py:18-38`;
        `tb_rollout_io.latent_sample_out`): z = mean where `latent_deterministic` [N, A] is set (or for every agent when `latent_eps`
        is None), mean + latent_eps * exp(log_std) elsewhere; the sample comes back as out["latent_sample"].
        `hidden_drop` [S] bool (host): train-mode `p_drop_hidden` with explicit draws -- the GRU state of all instances is zeroed after
        the steps where it is set (`waymo_motion.py:345-351`; `tb_rollout_io.hidden_drop`).
        `action_eps` [N, A, S, 2] standard normal: sampled actions (`deterministic_action=False`, `dynamics.py:77`); None = the mean.
        Closed-loop rollout of N = B*K instances (instance n uses scene n // K).  Returns the
        `RolloutBuffer` fields as [N, A, S, ...] tensors (`buffer.py:72-90`).  With `stepwise=True` only the
        prologue runs (`tb_rollout_begin`); advance with :meth:`rollout_step`, inspect with :meth:`rollout_state`.
        `gt` (:func:`gt_from_batch`) makes the full ground truth the source of initial state / overrides / the kill rule, as
        `reactive_replay` and the validation-time `joint_future_pred` do (`waymo_motion.py:457-461,538-545`); the traffic
        lights stay those of the history.  `latent_posterior` selects the posterior log_std for `latent_log_prob`.
        `warm_start_steps` = W > 0 promises that all valid agents are teacher-forced up to step W and none leaves before it
        (`tb_rollout_io.warm_start_steps`): same results, the first W+1 map / traffic-light attention halves run as one launch.
        With the default mask (mask_teacher_forcing=None) it is derived from the config and the scene's `warm_ok`."""
        n_tl_step = s["tl_valid"].shape[1]
        ag = s if gt is None else gt
        b, nh, a = ag["agent_valid"].shape
        p = s["map_valid"].shape[1]
        t = s["tl_valid"].shape[2]
        n = b * k_futures
        n_step = step_end - self.cfg["time_step_sim_start"] + 1
        dev, f32, u8 = self.device, torch.float32, torch.uint8
        if mask_teacher_forcing is None:
            tf = self.cfg["teacher_forcing_joint_future_pred"]
            if warm_start_steps == 0 and ag.get("warm_ok", False) and tf.get("step_warm_start", 10) >= 0:
                warm_start_steps = min(int(tf.get("step_warm_start", 10)), self.n_hist - 1)
            # (the default mask of a caller that re-uses its buffers -- `out=` -- is made once: same device address on every pass, so
            # tb_rollout can replay its captured graph; keyed by the validity tensor it was made from -- the object and its version)
            # tensor IDENTITY, not its address: the entry holds the validity tensor itself (a freed tensor's address can be handed to
            # the next batch by the caching allocator with the same version counter -- ADVICE r03)
            src = ag["agent_valid"]
            mkey = (src._version, tf.get("step_spawn_agent", 10), tf.get("step_warm_start", 10))
            cached = out.get("_default_tf_mask") if out is not None else None
            if cached is not None and cached[2] is src and cached[0] == mkey:
                mask_teacher_forcing = cached[1]
            elif ag.get("_tf_params") == (tf.get("step_spawn_agent", 10), tf.get("step_warm_start", 10)):
                mask_teacher_forcing = ag["_tf_mask"]  # made on the host when the batch was staged (staging.py)
            else:
                mask_teacher_forcing = as_u8(teacher_forcing_mask(
                    as_bool(ag["agent_valid"]), tf.get("step_spawn_agent", 10), tf.get("step_warm_start", 10)))
            default_mask = (mkey, mask_teacher_forcing, src)
        else:
            default_mask = None
        assert mask_teacher_forcing.shape == (b, nh, a)
        if out is None:
            out = {
                "preds": torch.empty(n, a, n_step, 4, device=dev, dtype=f32),
                "action_log_probs": torch.empty(n, a, n_step, device=dev, dtype=f32),
                "latent_log_prob": torch.empty(n, a, device=dev, dtype=f32),
                "final_state": torch.empty(n, a, 4, device=dev, dtype=f32),
                "final_valid": torch.empty(n, a, device=dev, dtype=u8),
                "final_hidden": torch.empty(3, n, a, 128, device=dev, dtype=f32),
            }
            for name in ("valid", "override_masks", "outside_map", "outside_map_this_step", "dest_reached",
                         "dest_reached_this_step"):
                out[name] = torch.empty(n, a, n_step, device=dev, dtype=u8)
            if record_actions:  # vis_dict["action"]: the physical action applied at every step
                out["actions"] = torch.zeros(n, a, n_step, 2, device=dev, dtype=f32)
            if tap_step >= 0 or tap_step == -2:  # (-2: every step, the buffers hold the latest)
                out["tap_policy_feature"] = torch.zeros(n, a, 128, device=dev, dtype=f32)
                out["tap_agent_feature"] = torch.zeros(n, a, 128, device=dev, dtype=f32)
            if record_check_states:  # what TrafficRuleChecker.check is handed every step (input of `rule_checks`)
                out["check_state"] = torch.zeros(n, a, n_step, 4, device=dev, dtype=f32)
                out["check_valid"] = torch.zeros(n, a, n_step, device=dev, dtype=u8)
        io = hip.TbRolloutIO()
        io.n_scene, io.k_futures, io.n_agent, io.n_pl, io.n_tl, io.n_hist, io.step_end = b, k_futures, a, p, t, nh, step_end
        io.map_feature = hip.ptr(feats["map_feature"], hip.c_f32p)
        io.map_feature_valid = hip.ptr(feats["map_feature_valid"], hip.c_u8p)
        io.tl_feature = hip.ptr(feats["tl_feature"], hip.c_f32p)
        io.tl_feature_valid = hip.ptr(s["tl_valid"], hip.c_u8p)
        io.n_tl_step, io.latent_posterior = n_tl_step, int(latent_posterior)
        io.warm_start_steps = 0 if stepwise else int(warm_start_steps)
        io.agent_valid = hip.ptr(ag["agent_valid"], hip.c_u8p)
        io.agent_state = hip.ptr(ag["agent_state"], hip.c_f32p)
        io.agent_vel = hip.ptr(ag["agent_vel"], hip.c_f32p)
        io.agent_acc = hip.ptr(ag["agent_acc"], hip.c_f32p)
        io.agent_yaw_rate = hip.ptr(ag["agent_yaw_rate"], hip.c_f32p)
        io.mask_teacher_forcing = hip.ptr(mask_teacher_forcing, hip.c_u8p)
        io.agent_type = hip.ptr(s["agent_type"], hip.c_i32p)
        io.agent_size = hip.ptr(s["agent_size"], hip.c_f32p)
        io.map_boundary = hip.ptr(s["map_boundary"], hip.c_f32p)
        io.map_valid = hip.ptr(s["map_valid"], hip.c_u8p)
        io.map_type = hip.ptr(s["map_type"], hip.c_i32p)
        io.map_pos = hip.ptr(s["map_pos"], hip.c_f32p)
        io.map_dir = hip.ptr(s["map_dir"], hip.c_f32p)
        latent_mean = latent_mean.to(f32).contiguous()
        dest = dest.to(torch.int32).contiguous()
        goal_valid = as_u8(goal_valid)
        assert dest.shape == (n, a) and goal_valid.shape == (n, a)
        assert latent_mean.shape == (b, a, 16)
        if latent_sample is None:  # drawn on the device by the prologue
            if latent_eps is not None:
                latent_eps = latent_eps.to(device=dev, dtype=f32).contiguous()
                assert latent_eps.shape == (n, a, 16), (tuple(latent_eps.shape), (n, a, 16))
            if latent_deterministic is not None:
                latent_deterministic = latent_deterministic.to(device=dev, dtype=u8).contiguous()
                assert latent_deterministic.shape == (n, a)
            if "latent_sample" not in out:
                out["latent_sample"] = torch.empty(n, a, 16, device=dev, dtype=f32)
            io.latent_eps = hip.ptr(latent_eps, hip.c_f32p)
            io.latent_deterministic = hip.ptr(latent_deterministic, hip.c_u8p)
            io.latent_sample_out = hip.ptr(out["latent_sample"], hip.c_f32p)
        else:
            latent_sample = latent_sample.to(f32).contiguous()
            assert latent_sample.shape == (n, a, 16)
            out["latent_sample"] = latent_sample
        io.latent_sample = hip.ptr(latent_sample, hip.c_f32p)
        io.latent_mean = hip.ptr(latent_mean, hip.c_f32p)
        io.dest = hip.ptr(dest, hip.c_i32p)
        io.goal_valid = hip.ptr(goal_valid, hip.c_u8p)
        io.preds = hip.ptr(out["preds"], hip.c_f32p)
        for name in ("valid", "override_masks", "outside_map", "outside_map_this_step", "dest_reached",
                     "dest_reached_this_step"):
            setattr(io, name, hip.ptr(out[name], hip.c_u8p))
        io.action_log_probs = hip.ptr(out["action_log_probs"], hip.c_f32p)
        io.latent_log_prob = hip.ptr(out["latent_log_prob"], hip.c_f32p)
        io.final_state = hip.ptr(out.get("final_state"), hip.c_f32p)
        io.final_valid = hip.ptr(out.get("final_valid"), hip.c_u8p)
        io.final_hidden = hip.ptr(out.get("final_hidden"), hip.c_f32p)
        io.tap_step = tap_step
        io.actions = hip.ptr(out.get("actions"), hip.c_f32p)
        io.tap_policy_feature = hip.ptr(out.get("tap_policy_feature"), hip.c_f32p)
        io.tap_agent_feature = hip.ptr(out.get("tap_agent_feature"), hip.c_f32p)
        io.check_state = hip.ptr(out.get("check_state"), hip.c_f32p)
        io.check_valid = hip.ptr(out.get("check_valid"), hip.c_u8p)
        if action_eps is not None:  # (fused or stepwise: the step kernel indexes the draws by the step it runs)
            action_eps = action_eps.to(device=dev, dtype=f32).contiguous()
            assert action_eps.shape == (n, a, n_step, 2), (tuple(action_eps.shape), (n, a, n_step, 2))
        io.action_eps = hip.ptr(action_eps, hip.c_f32p)
        hd = None
        if hidden_drop is not None:
            if stepwise:
                raise NotImplementedError("hidden_drop is built for the fused rollout, not for the stepwise API")
            hd = np.ascontiguousarray(np.asarray(torch.as_tensor(hidden_drop).cpu()).astype(np.uint8))
            assert hd.shape == (n_step,), (hd.shape, n_step)
            io.hidden_drop = C.c_void_p(hd.ctypes.data)
        # keep the borrowed inputs alive until the stream work is done
        if default_mask is not None:
            out["_default_tf_mask"] = default_mask
        out["_keepalive_host"] = hd
        out["_sampled_actions"] = action_eps is not None
        out["_keepalive"] = (latent_sample, latent_mean, dest, goal_valid, mask_teacher_forcing, s, feats, gt, action_eps, latent_eps,
                             latent_deterministic)
        if stepwise:
            self._check(self.lib.tb_rollout_begin(self._ctx, C.byref(io), self._stream()), "tb_rollout_begin")
            self._step_out = out
            self._step_open = True
        else:
            self._check(self.lib.tb_rollout(self._ctx, C.byref(io), self._stream()), "tb_rollout")
            self._step_open = False  # (tb_rollout reuses the stepwise context of the library)
        return out

    # -------------------------------------------------------------------------------- the samplers of joint_future_pred
    def latent_sample(self, mean: Tensor, k_futures: int = 1, eps: Optional[Tensor] = None, deterministic: Optional[Tensor] = None,
                      forced: Optional[Tensor] = None, posterior: bool = False, want_sample: bool = True, want_log_prob: bool = True,
                      log_std: Optional[Tensor] = None) -> Tuple[Optional[Tensor], Optional[Tensor]]:
        """`tb_latent_sample`: `MyDist.sample` / `DiagGaussian.log_prob` (`distributions.py:18-59`) of the personality distribution with
        the scene-level `mean` [B, A, 16] (instance n uses scene n // K); `log_std` [16] overrides the loaded prior / posterior parameter.
        Returns (sample [N, A, 16], log_prob [N, A])."""
        dev, f32 = self.device, torch.float32
        mean = mean.to(device=dev, dtype=f32).contiguous()
        b, a, _ = mean.shape
        n = b * k_futures
        io = hip.TbLatentSampleIO()
        io.n_scene, io.k_futures, io.n_agent, io.posterior = b, k_futures, a, int(posterior)
        eps = None if eps is None else eps.to(device=dev, dtype=f32).contiguous()
        det = None if deterministic is None else deterministic.to(device=dev, dtype=torch.uint8).contiguous()
        forced = None if forced is None else forced.to(device=dev, dtype=f32).contiguous()
        for t_, shape in ((eps, (n, a, 16)), (det, (n, a)), (forced, (n, a, 16))):
            assert t_ is None or tuple(t_.shape) == shape, (tuple(t_.shape), shape)
        sample = torch.empty(n, a, 16, device=dev, dtype=f32) if want_sample else None
        logp = torch.empty(n, a, device=dev, dtype=f32) if want_log_prob else None
        io.mean, io.eps, io.deterministic = hip.ptr(mean, hip.c_f32p), hip.ptr(eps, hip.c_f32p), hip.ptr(det, hip.c_u8p)
        io.forced, io.sample, io.log_prob = hip.ptr(forced, hip.c_f32p), hip.ptr(sample, hip.c_f32p), hip.ptr(logp, hip.c_f32p)
        if log_std is not None:
            log_std = log_std.to(device=dev, dtype=f32).expand(16).contiguous()
            io.log_std = hip.ptr(log_std, hip.c_f32p)
        self._check(self.lib.tb_latent_sample(self._ctx, C.byref(io), self._stream()), "tb_latent_sample")
        self._keep_sampler = (mean, eps, det, forced, log_std)
        return sample, logp

    def dest_sample(self, dest_logits: Tensor, k_futures: int = 1, uniform: Optional[Tensor] = None, deterministic: Optional[Tensor] = None,
                    forced: Optional[Tensor] = None, from_probs: bool = False, want_probs: bool = False
                    ) -> Tuple[Tensor, Tensor, Optional[Tensor]]:
        """`tb_dest_sample`: `DestCategorical.sample` / `log_prob` (`distributions.py:158-201`) on the masked destination logits
        [B, A, P]: arg max where deterministic (or for everybody when `uniform` is None), the inverse CDF of `uniform` [N, A] elsewhere;
        `forced` [N, A] scores given destinations instead.  Returns (sample [N, A] int32, log_prob [N, A], probs [B, A, P] or None)."""
        dev, f32 = self.device, torch.float32
        lg = dest_logits.to(device=dev, dtype=f32).contiguous()
        b, a, p = lg.shape
        n = b * k_futures
        io = hip.TbDestSampleIO()
        io.n_scene, io.k_futures, io.n_agent, io.n_pl, io.from_probs = b, k_futures, a, p, int(from_probs)
        u = None if uniform is None else uniform.to(device=dev, dtype=f32).contiguous()
        det = None if deterministic is None else deterministic.to(device=dev, dtype=torch.uint8).contiguous()
        forced = None if forced is None else forced.to(device=dev, dtype=torch.int32).contiguous()
        for t_ in (u, det, forced):
            assert t_ is None or tuple(t_.shape) == (n, a), (tuple(t_.shape), (n, a))
        sample = torch.empty(n, a, device=dev, dtype=torch.int32)
        logp = torch.empty(n, a, device=dev, dtype=f32)
        probs = torch.empty(b, a, p, device=dev, dtype=f32) if want_probs else None
        io.dest_logits, io.uniform, io.deterministic = hip.ptr(lg, hip.c_f32p), hip.ptr(u, hip.c_f32p), hip.ptr(det, hip.c_u8p)
        io.forced, io.sample, io.log_prob, io.probs = hip.ptr(forced, hip.c_i32p), hip.ptr(sample, hip.c_i32p), hip.ptr(logp, hip.c_f32p), hip.ptr(probs, hip.c_f32p)
        self._check(self.lib.tb_dest_sample(self._ctx, C.byref(io), self._stream()), "tb_dest_sample")
        self._keep_sampler = (lg, u, det, forced)
        return sample, logp, probs

    def rule_checks(self, s: Dict[str, Tensor], check_state: Tensor, check_valid: Tensor, k_futures: int,
                    flags: Dict[str, bool], tl: Optional[Dict[str, Tensor]] = None, agent_goal: Optional[Tensor] = None
                    ) -> Dict[str, Tensor]:
        """The flag-gated checks of `TrafficRuleChecker.check` (`traffic_rule_checker.py:122-335, 412-516`) over a recorded
        rollout (`tb_rule_checks`): `check_state` [N,A,S,4] / `check_valid` [N,A,S] as recorded by `rollout(...,
        record_check_states=True)`, `s` the pre-processed scene, `flags` the `traffic_rule_checker` config group.
        Returns the eight [N,A,S] uint8 arrays (zeros for a disabled check, like the reference).  `tl` (tl_valid / tl_state /
        tl_pos over any number of steps) replaces the history traffic lights (reactive_replay hands the checker the 91-step
        ground truth); `agent_goal` [B,A,4] adds goal_reached / goal_reached_this_step (`_check_goal_reached`, :337-361)."""
        n, a, n_step = check_valid.shape
        b = s["agent_valid"].shape[0]
        io = hip.TbRuleIO()
        io.n_scene, io.k_futures, io.n_agent, io.n_pl, io.n_tl, io.n_step = b, k_futures, a, s["map_valid"].shape[1], s["tl_valid"].shape[2], n_step
        io.enable_check_collided = int(bool(flags.get("enable_check_collided", False)))
        io.enable_check_run_road_edge = int(bool(flags.get("enable_check_run_road_edge", False)))
        io.enable_check_run_red_light = int(bool(flags.get("enable_check_run_red_light", False)))
        io.enable_check_passive = int(bool(flags.get("enable_check_passive", False)))
        cs, cv = check_state.to(torch.float32).contiguous(), check_valid.to(torch.uint8).contiguous()
        assert cs.shape == (n, a, n_step, 4) and n == b * k_futures
        io.check_state, io.check_valid = hip.ptr(cs, hip.c_f32p), hip.ptr(cv, hip.c_u8p)
        io.agent_type, io.agent_size = hip.ptr(s["agent_type"], hip.c_i32p), hip.ptr(s["agent_size"], hip.c_f32p)
        io.map_valid, io.map_type = hip.ptr(s["map_valid"], hip.c_u8p), hip.ptr(s["map_type"], hip.c_i32p)
        io.map_pos, io.map_dir = hip.ptr(s["map_pos"], hip.c_f32p), hip.ptr(s["map_dir"], hip.c_f32p)
        tls = s if tl is None else tl
        assert tls["tl_valid"].shape[2] == s["tl_valid"].shape[2]
        io.tl_valid, io.tl_state, io.tl_pos = hip.ptr(tls["tl_valid"], hip.c_u8p), hip.ptr(tls["tl_state"], hip.c_i32p), hip.ptr(tls["tl_pos"], hip.c_f32p)
        io.n_tl_step = tls["tl_valid"].shape[1]
        out = {}
        for name in RULE_KEYS:
            out[name] = torch.empty(n, a, n_step, device=self.device, dtype=torch.uint8)
            setattr(io, name, hip.ptr(out[name], hip.c_u8p))
        goal = None
        if agent_goal is not None:
            goal = agent_goal.to(self.device).to(torch.float32).contiguous()
            assert goal.shape == (b, a, 4)
            io.agent_goal = hip.ptr(goal, hip.c_f32p)
            for name in ("goal_reached", "goal_reached_this_step"):
                out[name] = torch.empty(n, a, n_step, device=self.device, dtype=torch.uint8)
                setattr(io, name, hip.ptr(out[name], hip.c_u8p))
        out["_keepalive"] = (cs, cv, s, tls, goal)
        self._check(self.lib.tb_rule_checks(self._ctx, C.byref(io), self._stream()), "tb_rule_checks")
        return out

    def metric_partials(self, pred_valid: Tensor, pred_states: Tensor, override_masks: Tensor, violations: Dict[str, Tensor],
                        agent_type: Tensor, agent_role: Tensor, gt_valid: Optional[Tensor] = None, gt_states: Optional[Tensor] = None,
                        loss_for_teacher_forcing: bool = False) -> Tensor:
        """The thirteen `dist_reduce_fx="sum"` states of the reference's `ErrorMetrics` / `TrafficRuleMetrics`
        (`src/models/metrics/logging.py:9-129`) for one rollout buffer, as a float64 device vector in `METRIC_FIELDS` order
        (`tb_metric_partials`) -- the packed partials the path's one collective sums over ranks.  Buffer tensors are
        [B,A,K,S(,4)], agent_type [B,A] index, agent_role [B,A,3], ground truth [B,A,S(,4)] or None."""
        dev, u8, f32 = self.device, torch.uint8, torch.float32
        b, a, k, n_step = pred_valid.shape
        io = hip.TbMetricIO()
        io.n_scene, io.n_agent, io.k_futures, io.n_step, io.loss_for_teacher_forcing = b, a, k, n_step, int(loss_for_teacher_forcing)
        keep = []

        def c8(t):
            t = t.to(dev).to(u8).contiguous()
            keep.append(t)
            return hip.ptr(t, hip.c_u8p)

        def cf(t):
            t = t.to(dev).to(f32).contiguous()
            keep.append(t)
            return hip.ptr(t, hip.c_f32p)

        io.pred_valid, io.pred_states, io.override_masks = c8(pred_valid), cf(pred_states), c8(override_masks)
        io.gt_valid = c8(gt_valid) if gt_valid is not None else hip.ptr(None, hip.c_u8p)
        io.gt_states = cf(gt_states) if gt_states is not None else hip.ptr(None, hip.c_f32p)
        io.agent_role = c8(agent_role)
        ty = agent_type.to(dev)
        ty = (ty.to(torch.int32).argmax(-1) if ty.dim() == 3 else ty).to(torch.int32).contiguous()
        keep.append(ty)
        io.agent_type = hip.ptr(ty, hip.c_i32p)
        zeros = torch.zeros(b, a, k, n_step, device=dev, dtype=u8)
        keep.append(zeros)
        for name in ("outside_map", "collided", "run_road_edge", "run_red_light", "passive", "goal_reached", "dest_reached"):
            setattr(io, name, c8(violations[name]) if name in violations else hip.ptr(zeros, hip.c_u8p))
        out = torch.zeros(len(METRIC_FIELDS), device=dev, dtype=torch.float64)
        io.out = C.cast(C.c_void_p(out.data_ptr()), C.POINTER(C.c_double))
        self._check(self.lib.tb_metric_partials(self._ctx, C.byref(io), self._stream()), "tb_metric_partials")
        self._metric_keepalive = keep
        return out

    def rollout_step(self, override: Optional[Dict[str, Tensor]] = None) -> None:
        """One simulation step of the rollout opened with `rollout(..., stepwise=True)` (`tb_rollout_step`).  `override`
        (`tb_rollout_step_ex`, per instance: "mask" [N,A], "agent_state" [N,A,4], "vel" [N,A,2], "acc" / "yaw_rate" [N,A(,1)],
        optional "gt_valid" [N,A]) replaces the history arrays as the teacher-forcing source of THIS step, the way the
        reference's `forward(state_override=, mask_state_override=)` + `Dynamics.kill(gt_valid)` do.  "action" [N,A,2] +
        "action_mask" [N,A] (the reference's `action_override / mask_action_override`, `dynamics.py:96-100`) replace the policy's
        physical action of this step for valid agents; they may come alone (no "mask": the bound teacher forcing applies)."""
        if override is None:
            self._check(self.lib.tb_rollout_step(self._ctx, self._stream()), "tb_rollout_step")
            return
        n, a = self._step_out["preds"].shape[:2]
        dev, f32, u8 = self.device, torch.float32, torch.uint8
        keep = {}

        def prep(key, dtype, shape):
            t = override[key].to(dev).to(dtype).reshape(shape).contiguous()
            keep[key] = t
            return t

        ov = hip.TbStepOverride()
        if override.get("mask") is not None:
            ov.mask = hip.ptr(prep("mask", u8, (n, a)), hip.c_u8p)
            ov.agent_state = hip.ptr(prep("agent_state", f32, (n, a, 4)), hip.c_f32p)
            ov.vel = hip.ptr(prep("vel", f32, (n, a, 2)), hip.c_f32p)
            ov.acc = hip.ptr(prep("acc", f32, (n, a)), hip.c_f32p)
            ov.yaw_rate = hip.ptr(prep("yaw_rate", f32, (n, a)), hip.c_f32p)
            ov.gt_valid = hip.ptr(prep("gt_valid", u8, (n, a)) if override.get("gt_valid") is not None else None, hip.c_u8p)
        if override.get("action_mask") is not None:
            ov.action = hip.ptr(prep("action", f32, (n, a, 2)), hip.c_f32p)
            ov.action_mask = hip.ptr(prep("action_mask", u8, (n, a)), hip.c_u8p)
        self._step_keepalive = keep  # borrowed until the stream work is done (replaced by the next step's)
        self._check(self.lib.tb_rollout_step_ex(self._ctx, C.byref(ov), self._stream()), "tb_rollout_step_ex")

    def forward_trunk(self, agent_valid: Tensor, agent_feature: Tensor, map_valid: Tensor, map_feature: Tensor, tl_valid: Tensor,
                      tl_feature: Tensor, goal_valid: Optional[Tensor], goal_feature: Optional[Tensor], latent_sample: Tensor,
                      hidden: Optional[Tensor], need_weights: bool = False) -> Dict[str, Tensor]:
        """`tb_forward`: the policy trunk of one step, un-fused, per instance (`TrafficBots.forward`, `traffic_bots.py:163-247`).
        `hidden` [3, N*A, 128] or None (= zeros); returns policy_feature [N,A,128], the new hidden [3, N*A, 128] and, with
        `need_weights`, the head-mean attention weights of the last layer of each block (attn_pl [N,A,P], attn_tl [N,A,T],
        attn_agent [N,A,A])."""
        dev, f32, u8 = self.device, torch.float32, torch.uint8
        n, a = agent_valid.shape
        p, t = map_valid.shape[1], tl_valid.shape[1]
        keep = []

        def c8(x):
            x = x.to(dev).to(u8).contiguous()
            keep.append(x)
            return hip.ptr(x, hip.c_u8p)

        def cf(x, shape):
            x = x.to(dev).to(f32).contiguous()
            assert tuple(x.shape) == tuple(shape), (tuple(x.shape), shape)
            keep.append(x)
            return hip.ptr(x, hip.c_f32p)

        io = hip.TbForwardIO()
        io.n_inst, io.n_agent, io.n_pl, io.n_tl = n, a, p, t
        io.agent_valid, io.agent_feature = c8(agent_valid), cf(agent_feature, (n, a, 128))
        io.map_valid, io.map_feature = c8(map_valid), cf(map_feature, (n, p, 128))
        io.tl_valid, io.tl_feature = c8(tl_valid), cf(tl_feature, (n, t, 128))
        if goal_feature is not None and goal_valid is not None:
            io.goal_valid, io.goal_feature = c8(goal_valid), cf(goal_feature, (n, a, 128))
        io.latent_sample = cf(latent_sample, (n, a, 16))
        h = torch.zeros(3, n * a, 128, device=dev, dtype=f32) if hidden is None else hidden.to(dev).to(f32).reshape(3, n * a, 128).clone()
        out = {"policy_feature": torch.empty(n, a, 128, device=dev, dtype=f32), "hidden": h}
        io.hidden, io.policy_feature = hip.ptr(h, hip.c_f32p), hip.ptr(out["policy_feature"], hip.c_f32p)
        if need_weights:
            out["attn_pl"] = torch.empty(n, a, p, device=dev, dtype=f32)
            out["attn_tl"] = torch.empty(n, a, t, device=dev, dtype=f32)
            out["attn_agent"] = torch.empty(n, a, a, device=dev, dtype=f32)
            io.attn_pl, io.attn_tl, io.attn_agent = (hip.ptr(out[k], hip.c_f32p) for k in ("attn_pl", "attn_tl", "attn_agent"))
        self._check(self.lib.tb_forward(self._ctx, C.byref(io), self._stream()), "tb_forward")
        self._fw_keepalive = keep
        return out

    def graph_stats(self) -> Dict[str, int]:
        """`tb_graph_stats`: rollouts captured as a hipGraph / replayed from one by this engine."""
        o = (C.c_int32 * 2)()
        self._check(self.lib.tb_graph_stats(self._ctx, o), "tb_graph_stats")
        return {"captured": int(o[0]), "replayed": int(o[1])}

    def check_status(self, raise_on_range: bool = True) -> bool:
        """`tb_check_status`: synchronises the current stream.  If an fp16-pair operand left the fp16 range since the last check the
        results of the calls in between are invalid and the CONTEXT HAS SWITCHED to the exact-fp32 kernels (include/trafficbots_hip.h):
        raises (default) or, with `raise_on_range=False`, returns True so that the caller can re-issue them.  Hard errors (a helper
        hand-off time-out) always raise."""
        rc = self.lib.tb_check_status(self._ctx, self._stream())
        if rc == 3:
            self._step_open = False  # the library closed an open stepwise rollout when it switched kernels (tb_api.hip)
        if rc == 3 and not raise_on_range:
            return True
        self._check(rc, "tb_check_status")
        return False

    def precision_state(self) -> Dict[str, object]:
        """`tb_precision_state`: the kernels this context runs on ("fp16_pair" / "bf16" / "fp32_exact") and why it left the configured
        ones (a loaded tensor outside the fp16-pair range / a run-time activation overflow)."""
        o = (C.c_int32 * 3)()
        self._check(self.lib.tb_precision_state(self._ctx, o), "tb_precision_state")
        names = {0: "fp16_pair", 1: "bf16", 2: "fp32_exact"}
        return {"step": names[int(o[0])], "encode": names[int(o[1])], "weight_out_of_range": bool(o[2] & 1),
                "activation_overflow": bool(o[2] & 2), "note": (self.lib.tb_precision_note(self._ctx) or b"").decode()}

    def precision_restore(self) -> bool:
        """`tb_precision_restore`: after a RUN-TIME fallback, back to the kernels `tb_finalize_weights` selected (no upload, no
        synchronisation); True when the selection changed.  A context whose loaded tensors are out of range stays where it is."""
        ch = C.c_int32(0)
        self._check(self.lib.tb_precision_restore(self._ctx, C.byref(ch)), "tb_precision_restore")
        if ch.value:
            self._step_open = False
        return bool(ch.value)

    def rollout_state(self) -> Dict[str, Tensor]:
        """Current simulator state of the stepwise rollout: `Dynamics.agent_state / agent_valid`, `TrafficBots.hidden`."""
        o = self._step_out
        n, a = o["preds"].shape[:2]
        st = {
            "agent_state": torch.empty(n, a, 4, device=self.device, dtype=torch.float32),
            "agent_valid": torch.empty(n, a, device=self.device, dtype=torch.uint8),
            "hidden": torch.empty(3, n, a, 128, device=self.device, dtype=torch.float32),
        }
        self._check(
            self.lib.tb_rollout_state(self._ctx, hip.ptr(st["agent_state"], hip.c_f32p), hip.ptr(st["agent_valid"], hip.c_u8p),
                                      hip.ptr(st["hidden"], hip.c_f32p), self._stream()), "tb_rollout_state")
        return st

    # -------------------------------------------------------------------------------- timing
    def set_timing(self, enable: bool) -> None:
        self._check(self.lib.tb_set_timing(self._ctx, int(enable)), "tb_set_timing")

    def get_timing(self) -> Dict[str, float]:
        buf = (C.c_float * 4)()
        self._check(self.lib.tb_get_timing(self._ctx, buf), "tb_get_timing")
        # fused launches = C(t)+A(t+1) (S-1 of them); edge launches = A(1) alone + C(S) alone
        return {"fused_ms": buf[0], "edge_ms": buf[1], "prologue_ms": buf[2], "n_fused": int(buf[3])}
