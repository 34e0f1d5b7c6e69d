"""Drop-in surface of the reference's task module for the hot path.

Same method names, argument meaning and outputs as `pl_modules.waymo_motion.WaymoMotion`
(`src/pl_modules/waymo_motion.py`) and `models.traffic_bots.TrafficBots`
(`src/models/traffic_bots.py`) for: `pre_processing`, `encode_input_features`,
`goal_manager.pred_goal` / `get_gt_goal`, `latent_encoder` (prior and posterior), `rollout`, `reactive_replay`,
`joint_future_pred`, `test_step`, `validation_step` (forward losses and metric states; no backward).
Everything numerical is executed by `libtrafficbots_hip.so` through :class:`HipEngine`; this file
is argument plumbing.  Lightning/Hydra are not required (and not rebuilt): the class is a plain
object that can be wrapped by a LightningModule in the reference harness.
"""
from __future__ import annotations

import collections
import weakref
from typing import Dict, Optional, Tuple, Union

import torch
from torch import Tensor

from .config import config_from_hydra_kwargs, load_model_config
from .distributions import DestCategorical, DiagGaussian
from .metrics import ErrorMetrics, TrafficRuleMetrics, TrainingMetrics
from .runtime import (HipEngine, as_bool, as_u8, gt_from_batch, hist_from_batch, history_len, scene_from_batch_torch, teacher_forcing_mask,
                      warm_start_promise_holds)

_VIOLATION_KEYS = (
    "outside_map", "outside_map_this_step", "collided", "collided_this_step", "run_road_edge",
    "run_road_edge_this_step", "run_red_light", "run_red_light_this_step", "passive", "passive_this_step",
    "goal_reached", "goal_reached_this_step", "dest_reached", "dest_reached_this_step",
)


_STAND_IN: Dict = {}  # device -> the one-element tensor behind the shape-only attr / pe stand-ins


class SceneDict(dict):
    """The pre-processed scene: a dict that can be weakly referenced (see `_with_reference_keys`)."""

    __slots__ = ("__weakref__",)


def _scene_of_stand_ins(tensors) -> Optional[Dict[str, Tensor]]:
    """The scene behind the shape-only attr / pe stand-ins of `_with_reference_keys`, None when `tensors` are caller-made."""
    for t in tensors:
        r = getattr(t, "_tb_scene", None)
        if r is not None:
            scene = r()
            if scene is None:
                raise RuntimeError("encode_input_features: these attr / pe tensors are the shape-only stand-ins of a pre-processed scene that "
                                   "no longer exists -- keep the dict `pre_processing` returned alive while its entries are in use")
            return scene
    return None


def retarget_stand_ins(scene) -> None:
    """After the entries of a pre-processed scene were moved into ANOTHER dict object (`staging.StagedBatch(scene)`): the stand-ins
    point (weakly) at that one from now on."""
    back = weakref.ref(scene)
    for v in scene.values():
        if torch.is_tensor(v) and getattr(v, "_tb_scene", None) is not None:
            v._tb_scene = back


def _with_reference_keys(scene: Dict[str, Tensor]) -> Dict[str, Tensor]:
    """Adds the entries the reference's harness reads off the pre-processed batch (`waymo_motion.py:902-921`): "input/*" and
    "latent_prior/*" (the twelve arguments of `encode_input_features`; in eval mode the second set aliases the first,
    `sc_latent.py:142-144,165-167,210-212`) and "ref/*" (`scene_centric.py:127-133`).  valid / pos entries are real tensors; the
    attr / pe entries are shape-only stand-ins (a one-element tensor expanded with stride 0: no memory, no arithmetic) that carry
    a reference to the scene, because the HIP encoders assemble the attributes and evaluate the pose PE from the raw scene inside
    `tb_encode_scene` (`sc_input.py:107-140`) -- `TrafficBots.encode_input_features` recognises them."""
    b, nh, a = scene["agent_valid"].shape
    p, t = scene["map_valid"].shape[1], scene["tl_valid"].shape[2]
    dev = scene["agent_valid"].device
    z = _STAND_IN.get(dev)
    if z is None:
        z = _STAND_IN[dev] = torch.zeros(1, device=dev, dtype=torch.float32)

    if type(scene) is dict:
        scene = SceneDict(scene)  # (a plain dict cannot be weakly referenced)
    back = weakref.ref(scene)

    def stand_in(*shape):
        x = z.expand(*shape)
        # a WEAK reference: scene -> stand-in -> scene would be a cycle, and cyclic garbage keeps the batch's device slab alive until
        # the cyclic collector happens to run -- its thresholds count objects, not bytes (tests/probes/gpu_soak.py: +0.3 MB per batch)
        x._tb_scene = back
        return x

    ref = {
        "agent_valid": as_bool(scene["agent_valid"]), "agent_attr": stand_in(b, nh, a, 11), "agent_pe": stand_in(b, nh, a, 96),
        "agent_pos": scene["agent_pos"],
        "map_valid": as_bool(scene["map_valid"]), "map_attr": stand_in(b, p, 20, 31), "map_pe": stand_in(b, p, 20, 96),
        "map_pos": scene["map_pos"][:, :, 0],
        "tl_valid": as_bool(scene["tl_valid"]), "tl_attr": stand_in(b, nh, t, 5), "tl_pe": stand_in(b, nh, t, 96), "tl_pos": scene["tl_pos"],
    }
    for k, v in ref.items():
        scene["input/" + k] = v
        scene["latent_prior/" + k] = v
    if "_ref_agent_type" in scene:  # a staged scene carries the batch's own one-hots (staging.py): nothing to rebuild on the device
        scene["ref/agent_type"], scene["ref/map_type"] = scene["_ref_agent_type"], scene["_ref_map_type"]
    else:
        idx = torch.arange(11, device=dev)
        scene["ref/agent_type"] = scene["agent_type"].unsqueeze(-1) == idx[:3]
        scene["ref/map_type"] = scene["map_type"].unsqueeze(-1) == idx
    scene["ref/agent_state"] = scene["agent_state"]
    return scene


def _scene_from_reference_inputs(agent_valid, agent_attr, agent_pe, agent_pos, map_valid, map_attr, map_pe, map_pos, tl_valid, tl_attr,
                                 tl_pe, tl_pos, device) -> Dict[str, Tensor]:
    """Scene for `tb_encode_scene` from tensors the CALLER produced with the reference's layout (`traffic_bots.py:125-137`,
    `sc_input.py:107-140`): attributes and pose PE go to the encoders as given (`tb_encode_io.ext_*`); the class indices the
    destination predictor masks with are the one-hot parts of the attributes (agent type: attr[..., 8:11] of history step 0, the
    attribute is the same at every step; polyline type: attr[..., :11] of node 0)."""
    f32, u8 = torch.float32, torch.uint8

    def f(x):
        return x.to(device).to(f32).contiguous()

    def onehot_idx(x):
        x = x.to(device)
        idx = x.argmax(-1).to(torch.int32)
        return torch.where(x.sum(-1) > 0.5, idx, torch.full_like(idx, -1)).contiguous()

    if agent_attr.shape[-1] != 11 or map_attr.shape[-1] != 31 or tl_attr.shape[-1] != 5:
        raise ValueError("attribute widths must be agent 11 / map 31 / tl 5 (sc_input.py:110-138)")
    if agent_pe.shape[-1] != 96 or map_pe.shape[-1] != 96 or tl_pe.shape[-1] != 96:
        raise ValueError("pose PE must be 96 wide (pe_xy_yaw with pe_dim 96)")
    s: Dict[str, Tensor] = {
        "agent_valid": agent_valid.to(device).to(u8).contiguous(), "map_valid": map_valid.to(device).to(u8).contiguous(),
        "tl_valid": tl_valid.to(device).to(u8).contiguous(),
        "agent_type": onehot_idx(agent_attr[:, 0, :, 8:11]), "map_type": onehot_idx(map_attr[:, :, 0, :11]),
        "ext_agent_attr": f(agent_attr), "ext_agent_pe": f(agent_pe), "ext_map_attr": f(map_attr), "ext_map_pe": f(map_pe),
        "ext_tl_attr": f(tl_attr), "ext_tl_pe": f(tl_pe),
        "agent_pos": f(agent_pos), "tl_pos": f(tl_pos), "warm_ok": False,
    }
    return s


class RolloutBuffer:
    """Fields of the reference buffer after `finish()` (`src/utils/buffer.py:72-90`)."""

    def __init__(self, step_start: int, step_end: int, step_current: int) -> None:
        self.step_start = step_start
        self.step_end = step_end
        self.step_future_start = step_current + 1 - step_start
        self.valid: Tensor = None  # [N, A, S] bool
        self.preds: Tensor = None  # [N, A, S, 4]
        self.override_masks: Tensor = None
        self.violations: Dict[str, Tensor] = {}
        self.latent_log_probs: Tensor = None
        self.action_log_probs: Tensor = None
        self.diffbar_rewards = []  # [N, A, S] once a rollout ran against ground truth (K = 1), else empty
        self.diffbar_rewards_valid = []
        self.vis_dicts: Dict[str, Tensor] = {}

    def flatten_repeat(self, n_repeat: int) -> None:
        """`buffer.py:92-123`: [B*K, A, S, ..] -> [B, A, K, S, ..]."""

        def fr(x: Tensor) -> Tensor:
            n = x.shape[0] // n_repeat
            return x.reshape(n, n_repeat, *x.shape[1:]).transpose(1, 2)

        self.valid = fr(self.valid)
        self.override_masks = fr(self.override_masks)
        self.preds = fr(self.preds)
        self.violations = {k: fr(v) for k, v in self.violations.items()}
        self.latent_log_probs = fr(self.latent_log_probs)
        self.action_log_probs = fr(self.action_log_probs)
        if torch.is_tensor(self.diffbar_rewards):
            self.diffbar_rewards, self.diffbar_rewards_valid = fr(self.diffbar_rewards), fr(self.diffbar_rewards_valid)
        self.vis_dicts = {k: fr(v) for k, v in self.vis_dicts.items()}  # (`buffer.py:118-123`)


class _GoalManager:
    goal_attr_mode = "dest"
    dummy = False
    update_goal = False

    def __init__(self, owner: "TrafficBots") -> None:
        self._o = owner

    def pred_goal(self, **kwargs) -> DestCategorical:
        """`GoalManager.pred_goal` -> `DestPredictor.forward` (`goal_manager.py:78-82,202-333`).  The logits
        were produced together with the features by `encode_input_features`."""
        return DestCategorical(logits=self._o._enc["dest_logits"], valid=self._o._goal_valid(), engine=self._o.engine)

    def get_gt_goal(self, agent_valid: Tensor, gt_goal: Optional[Tensor], gt_dest: Tensor) -> Tuple[Tensor, Tensor]:
        """`GoalManager.get_gt_goal`, goal_attr_mode "dest" (`goal_manager.py:50-75`)."""
        sc = self._o._scene
        if agent_valid is sc.get("agent_valid") or agent_valid is sc.get("input/agent_valid"):
            return gt_dest, self._o._goal_valid()
        return gt_dest, as_bool(agent_valid).any(1)


class TrafficBots:
    """Facade over the HIP encoders with the reference's method names (`traffic_bots.py:109-161`)."""

    def __init__(self, engine: HipEngine) -> None:
        self.engine = engine
        self.goal_manager = _GoalManager(self)
        self._enc: Dict[str, Tensor] = {}
        self._scene: Dict[str, Tensor] = {}
        self._log_std: Optional[Tensor] = None
        self._log_std_post: Optional[Tensor] = None

    def encode_input_features(self, agent_valid=None, agent_attr: Optional[Tensor] = None, agent_pe: Optional[Tensor] = None,
                              agent_pos: Optional[Tensor] = None, map_valid: Optional[Tensor] = None, map_attr: Optional[Tensor] = None,
                              map_pe: Optional[Tensor] = None, map_pos: Optional[Tensor] = None, tl_valid: Optional[Tensor] = None,
                              tl_attr: Optional[Tensor] = None, tl_pe: Optional[Tensor] = None, tl_pos: Optional[Tensor] = None
                              ) -> Dict[str, Tensor]:
        """`TrafficBots.encode_input_features` with the reference's argument list (`traffic_bots.py:109-151`):
        `model.encode_input_features(**input_dict)`, `input_dict` = the "input/*" (or "latent_prior/*") entries of the
        pre-processed batch.  One HIP call also yields the prior mean and the destination logits, which `latent_encoder` /
        `goal_manager.pred_goal` then hand out.  Three ways in:

        * the pre-processed scene of :meth:`WaymoMotion.pre_processing` as the only argument (the mirror's own short form);
        * the "input/*" entries of that scene: its attr / pe entries are shape-only stand-ins that point back at the scene (the
          kernels assemble the attributes and evaluate the pose PE themselves, `sc_input.py:107-140`), so this is the same call;
        * tensors the CALLER made (e.g. the reference's own `SceneCentricInput` outputs): attributes and PE are taken as given
          (`tb_encode_io.ext_*`), class indices for the destination predictor are read off the one-hot parts of the attributes.
        A second call on the same scene (the reference encodes "latent_prior/*" too, which aliases "input/*" in eval mode,
        `sc_latent.py:142-144`) returns the first call's result."""
        if isinstance(agent_valid, dict):
            scene = agent_valid
        else:
            given = (agent_valid, agent_attr, agent_pe, agent_pos, map_valid, map_attr, map_pe, map_pos, tl_valid, tl_attr, tl_pe, tl_pos)
            if any(t is None for t in given):
                raise TypeError("encode_input_features needs the pre-processed scene or all twelve tensors of the reference's signature")
            scene = _scene_of_stand_ins((agent_attr, agent_pe, map_attr, map_pe, tl_attr, tl_pe))
            if scene is None:
                scene = _scene_from_reference_inputs(agent_valid, agent_attr, agent_pe, agent_pos, map_valid, map_attr, map_pe, map_pos,
                                                     tl_valid, tl_attr, tl_pe, tl_pos, self.engine.device)
        if scene is not self._scene or not self._enc:
            self._enc = self.engine.encode_scene(scene)
            self._scene = scene
        e = self._enc
        return {
            "agent_feature": e["agent_feature"], "agent_feature_valid": as_bool(scene["agent_valid"]),
            "map_feature": e["map_feature"], "map_feature_valid": as_bool(e["map_feature_valid"]),
            "tl_feature": e["tl_feature"], "tl_feature_valid": as_bool(scene["tl_valid"]),
        }

    def _goal_valid(self) -> Tensor:
        """agent_valid.any(1) of the encoded scene [B, A] bool: made on the host for a staged scene (staging.py)."""
        sc = self._scene
        return as_bool(sc["_goal_valid"]) if "_goal_valid" in sc else as_bool(sc["agent_valid"]).any(1)

    def init(self, latent: DiagGaussian, deterministic: Union[bool, Tensor], eps: Optional[Tensor] = None) -> None:
        """`TrafficBots.init` (`traffic_bots.py:153-161`): binds the personality distribution for the next rollout and clears the
        recurrent state.  The sample itself is drawn when the simulator is opened (`WaymoMotion.rollout(..., latent=None)` picks it
        up); `eps` are the explicit standard-normal draws (the reference takes them from torch's global RNG)."""
        self.latent, self.deterministic, self._latent_eps = latent, deterministic, eps
        if latent is not None and latent.engine is None:
            latent.engine = self.engine  # (a distribution the caller made by hand is served by this model's device engine)
        self.hidden = None
        self.latent_sample = None
        self.latent_logp = None

    def forward(self, agent_valid: Tensor, agent_feature: Tensor, map_valid: Tensor, map_feature: Tensor, tl_valid: Tensor,
                tl_feature: Tensor, goal_valid: Optional[Tensor], goal_feature: Optional[Tensor], need_weights: bool = False):
        """`TrafficBots.forward` (`traffic_bots.py:163-247`) with the reference's signature and return value
        `(policy_feature, latent_logp, attn_pl, attn_tl, attn_agent)`: one step of the policy trunk on per-instance tensors, un-fused
        (`tb_forward`).  Stateful like the reference: the personality is sampled on the first call after `init(latent, deterministic)`
        (`:196-199`; explicit `eps` if `init` got them) and `self.hidden` carries the GRU state from call to call.  With
        `need_weights=True` the attention weights are the head-mean weights of the last layer of each block (None otherwise).
        The rollout does NOT go through here -- its trunk is fused into the step kernel (`WaymoMotion.forward` / `rollout`); this is the
        visualisation / debugging entry point and an on-device cross-check of that kernel."""
        if getattr(self, "latent", None) is None:
            raise RuntimeError("TrafficBots.forward: call init(latent, deterministic) first")
        if self.latent_sample is None:
            self.latent_sample = self.latent.sample(self.deterministic, eps=self._latent_eps)
            self.latent_logp = self.latent.log_prob(self.latent_sample)
        out = self.engine.forward_trunk(agent_valid, agent_feature, map_valid, map_feature, tl_valid, tl_feature, goal_valid, goal_feature,
                                        self.latent_sample, self.hidden, need_weights=need_weights)
        self.hidden = out["hidden"]
        return out["policy_feature"], self.latent_logp, out.get("attn_pl"), out.get("attn_tl"), out.get("attn_agent")

    __call__ = forward

    def latent_encoder(self, posterior: bool = False, gt: Optional[Dict[str, Tensor]] = None, **kwargs) -> DiagGaussian:
        """`LatentEncoder.forward` (`latent_encoder.py:70-147`).  The prior was produced by `encode_input_features`; the
        posterior (`posterior=True`) runs `tb_encode_posterior` on the full ground truth `gt` (the "gt" entry of the
        pre-processed validation scene) and the map feature of the last `encode_input_features`."""
        if posterior:
            if gt is None:
                raise ValueError("latent_encoder(posterior=True) needs the ground truth (pre_processing(batch)['gt'])")
            post = self.engine.encode_posterior(gt, self._enc)
            return DiagGaussian(post["latent_mean"], self._log_std_post, valid=as_bool(post["latent_valid"]), engine=self.engine)
        return DiagGaussian(self._enc["latent_mean"], self._log_std, valid=as_bool(self._enc["latent_valid"]), engine=self.engine)


def _range_fallback(fn):
    """A harness step on the fp16-pair kernels whose activations left their range (|x| >= 65504) is RE-RUN on the exact-fp32 kernels
    instead of failing: `tb_check_status` reports the overflow, the context has then switched itself (fp32 MFMA kernels, fp32's range)
    and the step is issued again -- the reference has no range limit (`src/models/modules/mlp.py:20-85`), so a checkpoint that trips
    the guard still gets an fp32-accurate result, only slower (the warning states the measured slowdown).  Metric holders are restored
    to their state before the invalid run, and so is the state of a `generator` the step draws from: the re-run sees the SAME random
    numbers as the invalid run did (round 5; before, it silently drew new ones).  `check_range = False` skips the check (and the
    fallback)."""
    import copy
    import functools
    import inspect
    import time
    import warnings

    sig = inspect.signature(fn)

    @functools.wraps(fn)
    def step(self, *args, **kwargs):
        if not self.check_range:
            return fn(self, *args, **kwargs)
        # (the holders' states are replaced, never modified in place -- metrics._PackedSumMetric.update: a shallow copy restores them)
        snap = [copy.copy(h.__dict__) for h in self._metric_holders()] if fn.__name__ != "test_step" else []
        # the generator wherever it was passed (keyword or positional: ADVICE r05); with none, the draws come from torch's default
        # generators, whose states are snapshotted instead
        try:
            gen = sig.bind(self, *args, **kwargs).arguments.get("generator")
        except TypeError:
            gen = kwargs.get("generator")
        gen_state = gen.get_state() if gen is not None else None
        default_state = None if gen is not None else (torch.random.get_rng_state(), torch.cuda.get_rng_state(self.device))
        t0 = time.perf_counter()
        out = fn(self, *args, **kwargs)
        if not self.engine.check_status(raise_on_range=False):
            self._fallback_hist.append(0)
            return out
        t1 = time.perf_counter()
        for h, d in zip(self._metric_holders(), snap):
            h.__dict__.clear()
            h.__dict__.update(d)
        if gen_state is not None:
            gen.set_state(gen_state)
        elif default_state is not None:
            torch.random.set_rng_state(default_state[0])
            torch.cuda.set_rng_state(default_state[1], self.device)
        out = fn(self, *args, **kwargs)
        if self.engine.check_status(raise_on_range=False):
            # only the kernel family that overflowed had switched (say the scene encoders), and with its results now finite the other
            # one overflowed in turn: one more run, everything that can raise the flag is on its exact twin by now
            for h, d in zip(self._metric_holders(), snap):
                h.__dict__.clear()
                h.__dict__.update(d)
            if gen_state is not None:
                gen.set_state(gen_state)
            elif default_state is not None:
                torch.random.set_rng_state(default_state[0])
                torch.cuda.set_rng_state(default_state[1], self.device)
            out = fn(self, *args, **kwargs)
            self.engine.check_status()  # (the exact kernels cannot raise the flag; anything else is a hard error)
        t2 = time.perf_counter()
        note = self.engine.precision_state()["note"]
        warnings.warn(f"trafficbots_amd: {note} -- {fn.__name__} was re-run on the exact-fp32 kernels: "
                      f"{(t2 - t1) * 1e3:.1f} ms against {(t1 - t0) * 1e3:.1f} ms for the invalid fp16-pair run ({(t2 - t1) / max(t1 - t0, 1e-9):.2f}x); "
                      + self._after_fallback(), RuntimeWarning, stacklevel=2)
        return out

    return step


class WaymoMotion:
    def __init__(self, config_path: Optional[str] = None, device: str = "cuda:0", **overrides) -> None:
        """Two call forms.  The mirror's own: `WaymoMotion(config_path=None, device=..., **dotted_overrides)`.  The reference's:
        `WaymoMotion(time_step_current=..., time_step_gt=..., ..., model={...}, dynamics={...}, ...)` -- the keyword arguments
        `hydra.utils.instantiate` passes for `configs/model/traffic_bots.yaml` (`waymo_motion.py:28-62`); nested groups as plain
        dicts (or anything dict-like), `_target_` keys and the training-only groups (optimizer, lr_scheduler, sub_womd_*, data_size,
        wb_artifact ...) accepted and not used.  See also :func:`trafficbots_amd.instantiate`."""
        self._ctor = (config_path, device, dict(overrides))  # (for `clone`: a second context with the same configuration)
        self._state_dict = None
        self._lane_clones: list = []   # the other contexts `pipeline` runs on: made once, kept (a context is an arena + workspaces)
        self._lane_streams: list = []
        if isinstance(overrides.get("model"), dict) or hasattr(overrides.get("model"), "items"):
            self.hparams = config_from_hydra_kwargs(overrides)
        else:
            self.hparams = load_model_config(config_path, overrides or None)
        self.check_range = True  # tb_check_status at the end of test_step / validation_step / training_step (one stream sync each)
        # What a context does AFTER a batch overflowed the fp16-pair range and was re-run on the exact-fp32 kernels (`_range_fallback`):
        # "adaptive" (default) goes back to the fast kernels for the next batch (`tb_precision_restore`: one outlier scene costs one
        # re-run, not half the throughput of everything after it) unless `fallback_sticky_after` of the last `fallback_window` checked
        # steps overflowed -- a checkpoint that overflows systematically stays on the exact kernels instead of paying both runs per
        # batch; "sticky": stay after the first overflow (the behaviour up to round 5); "per_batch": always go back.
        self.fallback_policy = "adaptive"
        self.fallback_sticky_after, self.fallback_window = 3, 16
        self._fallback_hist = collections.deque(maxlen=self.fallback_window)
        self.n_fallbacks = 0
        self._det_cache: Dict = {}
        self._zeros_cache: Dict = {}
        self.device = torch.device(device)
        self.engine = HipEngine(self.hparams, device)
        self.model = TrafficBots(self.engine)
        self.n_hist = self.hparams["time_step_current"] + 1
        from .post_processing import WaymoPostProcessing

        self.waymo_post_processing = WaymoPostProcessing(self.engine, **self.hparams.get("waymo_post_processing", {}))
        # metric holders of validation_step (waymo_motion.py:86-105); the states are produced on the GPU
        tm = self.hparams["training_metrics"]
        self.train_metrics_reactive_replay = TrainingMetrics("reactive_replay", **tm)
        self.err_metrics_reactive_replay = ErrorMetrics("reactive_replay")
        self.rule_metrics_reactive_replay = TrafficRuleMetrics("reactive_replay")
        self.err_metrics_joint_future_pred = ErrorMetrics("joint_future_pred")
        self.rule_metrics_joint_future_pred = TrafficRuleMetrics("joint_future_pred")

    # ------------------------------------------------------------------ weights
    def load_state_dict(self, state_dict: Dict) -> None:
        self._state_dict = state_dict
        self.engine.load_state_dict(state_dict)
        for c in getattr(self, "_lane_clones", []):  # (the other lanes of `pipeline` follow: same weights on every context)
            c.load_state_dict(state_dict)
        ls = state_dict["model.latent_encoder.latent_prior_dist.log_std"]
        self.model._log_std = torch.as_tensor(ls, dtype=torch.float32).to(self.device)
        lp = state_dict["model.latent_encoder.latent_post_dist.log_std"]
        self.model._log_std_post = torch.as_tensor(lp, dtype=torch.float32).to(self.device)

    # ------------------------------------------------------------------ pre-processing
    def pre_processing(self, batch: Dict[str, Tensor]) -> Dict[str, Tensor]:
        """Eval-mode `SceneCentricPreProcessing` + the layout half of `SceneCentricInput`
        (`scene_centric.py:103-133`, `sc_input.py:100-140`); attr/PE/MLP run inside `tb_encode_scene`.  A validation /
        training batch (one that carries "agent/valid") also yields the full ground truth as the nested dict `scene["gt"]`
        (`scene_centric.py:103-110`, `sc_latent.py:150-163,196-217`)."""
        from .staging import StagedBatch, is_host_batch

        if not isinstance(batch, StagedBatch) and "packed/agent_valid" not in batch and history_len(batch) > self.n_hist:
            # time_step_current below the batch's history length: the rollout of test_step still sees the whole history
            # (runtime.hist_from_batch); the scene itself is made from the batch as always
            hist = hist_from_batch(batch, self.device, self.n_hist, self._tf_params)
            scene = self._pre_processing(batch)
            scene["hist"] = hist
            return scene
        return self._pre_processing(batch)

    def _pre_processing(self, batch: Dict[str, Tensor]) -> Dict[str, Tensor]:
        from .staging import StagedBatch, is_host_batch

        if isinstance(batch, StagedBatch):  # staged (and possibly encoded) ahead of time by `prefetch`: this stream waits for it
            batch.wait(torch.cuda.current_stream(self.device))
            if batch.enc is not None:
                self.model._enc, self.model._scene = batch.enc, batch
            return batch
        if "packed/agent_valid" in batch:  # a `data_h5.PackedSceneLoader` batch: decoded by the reader, upload only
            from .data_h5 import scene_from_packed

            if int(batch["packed/agent_valid"].shape[1]) != self.n_hist or self.n_hist != 11:
                # (the packed reader decodes the first n_hist history steps only; test_step's rollout needs all eleven: hist_from_batch)
                raise NotImplementedError("trafficbots_amd: packed-h5 batches are built for time_step_current = 10 (the history the files "
                                          "carry); with another value feed the reference-layout batch (PackedH5File.read_reference_batch)")
            return _with_reference_keys(scene_from_packed(batch, self.device, self.n_hist, self._tf_params))
        if is_host_batch(batch):  # what a DataLoader yields: ONE pinned slab, one copy, no conversion kernels (staging.py)
            return _with_reference_keys(self.engine.stager(self._tf_params).stage(batch))
        scene = scene_from_batch_torch(batch, self.device, self.n_hist)  # (a batch somebody moved to the device already)
        if "agent/valid" in batch:
            scene["gt"] = gt_from_batch(batch, self.device, self.n_hist)
        return _with_reference_keys(scene)

    def clone(self) -> "WaymoMotion":
        """A second, independent context (own `tb_ctx`: weights arena, workspaces, status word) with this object's configuration and
        weights -- what `pipeline` runs its other lanes on."""
        config_path, device, overrides = self._ctor
        w = WaymoMotion(config_path, device, **overrides)
        if self._state_dict is not None:
            w.load_state_dict(self._state_dict)
        w.check_range = self.check_range
        w.fallback_policy, w.fallback_sticky_after, w.fallback_window = self.fallback_policy, self.fallback_sticky_after, self.fallback_window
        return w

    def _after_fallback(self) -> str:
        """Policy step after a batch was re-run on the exact-fp32 kernels (see `fallback_policy`); returns the clause for the warning."""
        if self._fallback_hist.maxlen != self.fallback_window:
            self._fallback_hist = collections.deque(self._fallback_hist, maxlen=self.fallback_window)
        self._fallback_hist.append(1)
        self.n_fallbacks += 1
        if self.fallback_policy not in ("adaptive", "sticky", "per_batch"):
            raise ValueError(f"fallback_policy = {self.fallback_policy!r}: expected 'adaptive', 'sticky' or 'per_batch'")
        recent = sum(self._fallback_hist)
        stay = self.fallback_policy == "sticky" or (self.fallback_policy == "adaptive" and recent >= self.fallback_sticky_after)
        if not stay and self.engine.precision_restore():
            return "the context is back on the fp16-pair kernels for the next batch (fallback_policy = %r)" % self.fallback_policy
        if self.fallback_policy == "adaptive" and stay:
            return f"this context stays on them ({recent} of the last {len(self._fallback_hist)} checked steps overflowed)"
        return "this context stays on them"

    def pipeline(self, loader, lanes: int = 3, step: str = "test_step", kwargs_fn=None):
        """`for out in wm.pipeline(loader): ...`: the harness step of consecutive batches on `lanes` contexts / streams, results in
        order and range-checked (`staging.LanePipeline`): two 32-scene rollouts in flight fill the chip that one leaves half empty, a
        third covers the gaps of the other two (their encoders, prologues, range checks).  Measured over five bench runs at the headline
        shape: three lanes 5.29 - 5.38 ms per batch, two lanes 5.49 - 6.43 (the two fall in and out of step); the third context costs
        ~0.4 GB.  K = 6: 15.6 vs 17.0 ms."""
        from .staging import LanePipeline

        return LanePipeline(self, loader, lanes=lanes, step=step, kwargs_fn=kwargs_fn)

    def _lanes(self, n: int):
        """The `n` contexts and streams `pipeline` uses: this object + clones that are made ONCE and kept (a clone is a weights arena,
        two workspaces and a host-side packing of every tensor -- hundreds of megabytes and a second of host time: a pipeline per
        epoch must not pay, or leak, that each time; tests/probes/gpu_soak.py).  Policy attributes follow this object at every call."""
        while len(self._lane_clones) < n - 1:
            self._lane_clones.append(self.clone())
        while len(self._lane_streams) < n:
            self._lane_streams.append(torch.cuda.Stream(device=self.device))
        for c in self._lane_clones:
            c.check_range = self.check_range
            c.fallback_policy, c.fallback_sticky_after, c.fallback_window = self.fallback_policy, self.fallback_sticky_after, self.fallback_window
        return [self] + self._lane_clones[: n - 1], self._lane_streams[:n]

    def prefetch(self, loader, encode: bool = True):
        """`for staged in wm.prefetch(loader): out = wm.test_step(staged)` (also validation_step): batch n + 1 is staged -- host packing,
        ONE upload -- and, with `encode`, run through the scene encoders on a side stream while batch n's rollout occupies the main
        one (`staging.BatchPrefetcher`).  Same results as `test_step(batch)`, bit for bit."""
        from .staging import BatchPrefetcher

        return BatchPrefetcher(self, loader, encode=encode)

    @property
    def _tf_params(self) -> Tuple[int, int]:
        tf = self.hparams["teacher_forcing_joint_future_pred"]
        return int(tf.get("step_spawn_agent", 10)), int(tf.get("step_warm_start", 10))

    # ------------------------------------------------------------------ rollout
    def rollout(
        self,
        features: Dict[str, Tensor],
        latent: Optional[DiagGaussian],
        goal: Tensor,
        goal_valid: Tensor,
        mask_teacher_forcing: Tensor,
        rule_checker=None,
        deterministic_latent: Union[bool, Tensor] = True,
        deterministic_action: bool = True,
        step_end: int = 90,
        step_start: int = 1,
        require_vis_dict: bool = False,
        gt_sdc=None,
        k_futures: int = 1,
        latent_eps: Optional[Tensor] = None,
        tap_step: int = -1,
        stepwise: bool = False,
        gt: Optional[Dict[str, Tensor]] = None,
        latent_posterior: bool = False,
        rule_checker_tl: Optional[Dict[str, Tensor]] = None,
        warm_start_steps: int = 0,
        action_eps: Optional[Tensor] = None,
        generator=None,
        hidden_drop=None,
    ) -> RolloutBuffer:
        """`hidden_drop` [S] bool: the train-mode hidden-state drop with explicit draws (`waymo_motion.py:345-351`), fused rollout only.
        `deterministic_action=False` samples every step's action (`dynamics.py:77`): the standard-normal draws are `action_eps`
        [N, A, S, 2] (explicit, as the goldens pass them) or, when None, drawn here with `torch.randn(generator=generator)` -- the
        reference draws them step by step from torch's global stream, which no other implementation can replay.
        `WaymoMotion.rollout` (`waymo_motion.py:205-354`).  `gt` (the scene's "gt" dict) replaces the history as
        `features["agent_valid" / "agent_state" / ...]`, the way validation and training call it (`:457-461`); with it and
        K = 1 the per-step `DifferentiableReward` is attached to the buffer (`:320-330`).  `stepwise=True` only initialises the simulator
        (`model.init`, `dynamics.init`, goal features: `waymo_motion.py:246-266`); drive it with :meth:`forward`.  `features` is the pre-processed scene merged
        with the encoder outputs (un-repeated: K futures share scene tensors, instance n uses scene n // K);
        `latent` / `goal` / `goal_valid` are per instance [N, ...] as in the reference."""
        if require_vis_dict and (stepwise or not deterministic_action or hidden_drop is not None):
            raise NotImplementedError("require_vis_dict drives the loop per step itself: not with stepwise=True / sampled actions / hidden_drop")
        if latent is None:  # the reference's `self.model.init(latent, deterministic)` + rollout without re-passing them
            latent, deterministic_latent = self.model.latent, self.model.deterministic
            latent_eps = latent_eps if latent_eps is not None else getattr(self.model, "_latent_eps", None)
        # what-if (the SDC's trajectory is forced at every step) and require_vis_dict (every step's attention weights come from the
        # un-fused tb_forward, `:167,191-201`) drive the loop per step
        per_step = gt_sdc is not None or require_vis_dict
        if deterministic_action:
            action_eps = None
        else:
            n_inst, n_ag = goal.shape[0], goal.shape[1]
            if action_eps is None:
                action_eps = torch.randn(n_inst, n_ag, step_end - step_start + 1, 2, device=self.device, generator=generator)
        if step_start != self.hparams["time_step_sim_start"]:
            raise NotImplementedError("step_start must equal time_step_sim_start")
        # the personality is drawn by the rollout prologue (MyDist.sample, distributions.py:18-38; tb_rollout_io.latent_sample_out)
        z_eps, z_det = latent._draws(deterministic_latent, latent_eps, generator)
        b = features["agent_valid"].shape[0]
        if warm_start_steps > 0 and not warm_start_promise_holds(
                mask_teacher_forcing, (gt if gt is not None else features)["agent_valid"], warm_start_steps):
            warm_start_steps = 0  # a caller-made mask that does not force every valid agent: step by step
        if latent._k == k_futures:
            mean_scene = latent._mean_scene
        else:  # a per-instance distribution made by hand: the K instances of a scene share its mean (repeat_interleave_, :493)
            mean_scene = latent.mean.reshape(b, k_futures, *latent.mean.shape[1:])[:, 0].contiguous()
        # traffic_rule_checker.enable_check_* (traffic_bots.yaml:240-244): the flag-gated checks do not feed back into the
        # simulation, they are evaluated on the recorded per-step states once the rollout is enqueued (tb_rule_checks)
        flags = self.hparams.get("traffic_rule_checker", {})
        # (and goal_reached whenever the batch carries a ground-truth goal, traffic_rule_checker.py:473-479)
        agent_goal = gt.get("gt_goal") if gt is not None else None
        want_checks = any(bool(v) for k, v in flags.items() if k.startswith("enable_check_")) or agent_goal is not None
        out = self.engine.rollout(
            features, features, None, mean_scene, goal, goal_valid, k_futures, step_end, latent_eps=z_eps, latent_deterministic=z_det,
            mask_teacher_forcing=as_u8(mask_teacher_forcing), tap_step=-2 if require_vis_dict else tap_step,
            stepwise=stepwise or per_step,
            record_check_states=want_checks, gt=gt, latent_posterior=latent_posterior, warm_start_steps=warm_start_steps,
            action_eps=action_eps, hidden_drop=hidden_drop, record_actions=require_vis_dict,
        )
        self._vis = None
        if require_vis_dict:  # what `forward(require_vis_dict=True)` needs from step to step
            rep = (lambda x: x.repeat_interleave(k_futures, 0)) if k_futures > 1 else (lambda x: x)
            map_f = rep(features["map_feature"])
            self._vis = {"k": k_futures, "z": out["latent_sample"], "goal_valid": goal_valid.to(self.device).bool().clone(),
                         "map_feature": map_f, "map_valid": rep(features["map_feature_valid"].bool()),
                         "goal_feature": torch.gather(map_f, 1, goal.to(self.device).long().clamp(0, map_f.shape[1] - 1).unsqueeze(-1).expand(-1, -1, 128)),
                         "tl_feature": features["tl_feature"], "tl_valid": features["tl_valid"].bool(), "rep": rep, "collected": []}
        if per_step:
            self._step_t = step_start
            self._rollout_with_per_step_overrides(features, mask_teacher_forcing, k_futures, step_start, step_end, gt, gt_sdc,
                                                  require_vis_dict=require_vis_dict)
            st = self.engine.rollout_state()
            out["final_state"], out["final_valid"], out["final_hidden"] = st["agent_state"], st["agent_valid"], st["hidden"]
        if (gt is not None and k_futures == 1 and not stepwise and self.hparams["training_metrics"]["w_diffbar_reward"] > 0
                and gt["agent_valid"].shape[1] > step_end):
            gv, gs = self._gt_slices(gt, step_start, step_end)
            out["diffbar_rewards"], out["diffbar_rewards_valid"], _ = self.engine.train_partials(out, gv, gs, features["agent_size"])
        z = out["latent_sample"]
        self._step_t = step_start
        self._rollout_meta = (step_start, step_end, z)
        self._rule_ctx = (features, k_futures, flags, rule_checker_tl, agent_goal) if want_checks else None
        self.model.latent_sample = z
        if want_checks and not stepwise:
            out.update(self.engine.rule_checks(features, out["check_state"], out["check_valid"], k_futures, flags,
                                               tl=rule_checker_tl, agent_goal=agent_goal))
        return self._buffer_from(out)

    def _buffer_from(self, out: Dict[str, Tensor]) -> RolloutBuffer:
        step_start, step_end, z = self._rollout_meta
        buf = RolloutBuffer(step_start, step_end, self.hparams["time_step_current"])
        buf.valid = as_bool(out["valid"])  # (the kernels write 0 / 1: a reinterpreting view, not a conversion)
        buf.preds = out["preds"]
        buf.override_masks = as_bool(out["override_masks"])
        zeros = None
        if any(k not in out for k in _VIOLATION_KEYS):  # (checks that did not run: all-False, one shared tensor)
            zk = (tuple(buf.valid.shape), buf.valid.device)
            zeros = self._zeros_cache.get(zk)
            if zeros is None:
                self._zeros_cache.clear()
                zeros = self._zeros_cache[zk] = torch.zeros_like(buf.valid)
        buf.violations = {k: (as_bool(out[k]) if k in out else zeros) for k in _VIOLATION_KEYS}
        n_step = buf.valid.shape[2]
        buf.latent_log_probs = out["latent_log_prob"].unsqueeze(-1).expand(-1, -1, n_step)
        buf.action_log_probs = out["action_log_probs"]
        buf.final = {k: out[k] for k in ("final_state", "final_valid", "final_hidden")}
        buf.taps = {k: out[k] for k in ("tap_policy_feature", "tap_agent_feature") if k in out}
        buf.latent_sample = z
        vis = getattr(self, "_vis", None)
        if vis is not None and vis["collected"]:  # `RolloutBuffer.finish`: torch.stack(..., dim=2) (`buffer.py:89-90`)
            buf.vis_dicts = {k_: torch.stack([d[k_] for d in vis["collected"]], dim=2) for k_ in vis["collected"][0]}
        if "diffbar_rewards" in out:
            buf.diffbar_rewards, buf.diffbar_rewards_valid = out["diffbar_rewards"], as_bool(out["diffbar_rewards_valid"])
        return buf

    def _warm_start_steps(self, tf: Dict, src: Dict[str, Tensor]) -> int:
        """W for `tb_rollout_io.warm_start_steps`: the TeacherForcing warm start (all valid agents forced up to step_warm_start,
        `teacher_forcing.py:56-58`) when no agent leaves within it (`warm_ok`, computed on the host batch), else 0."""
        if not src.get("warm_ok", False) or tf.get("step_warm_start", 10) < 0:
            return 0
        return min(int(tf.get("step_warm_start", 10)), self.n_hist - 1)

    @staticmethod
    def _gt_slices(gt: Dict[str, Tensor], step_start: int, step_end: int) -> Tuple[Tensor, Tensor]:
        """batch["gt/valid"][:, s0:].transpose(1, 2), batch["gt/state"][:, s0:].transpose(1, 2) (`waymo_motion.py:616-617`)."""
        gv = gt["agent_valid"][:, step_start: step_end + 1].transpose(1, 2).contiguous()
        gs = gt["agent_state"][:, step_start: step_end + 1].transpose(1, 2).contiguous()
        return gv, gs

    def reactive_replay(self, batch: Dict[str, Tensor], input_feature_dict: Dict[str, Tensor], mask_teacher_forcing: Tensor,
                        latent: DiagGaussian, goal: Tensor, goal_valid: Tensor, deterministic_latent: bool = True,
                        deterministic_action: bool = True, require_vis_dict: bool = False, latent_eps: Optional[Tensor] = None,
                        latent_is_posterior: bool = True, teacher_forcing_cfg: Optional[Dict] = None,
                        action_eps: Optional[Tensor] = None, generator=None, hidden_drop=None) -> RolloutBuffer:
        """`WaymoMotion.reactive_replay` (`waymo_motion.py:420-476`): the episode replayed from its ground truth (`batch` is the
        pre-processed validation scene, `batch["gt"]` its ground truth) with the given personality and goal; K = 1.  The
        personality is taken as a POSTERIOR for `latent_log_prob` (that is what validation / training pass, `:382-387,605`).
        `deterministic_latent=False` draws the personality as mean + std * `latent_eps` (training_step's rsample, `:397`)."""
        features = {k: v for k, v in batch.items() if k != "gt"}
        features["map_feature"] = input_feature_dict["map_feature"]
        features["map_feature_valid"] = as_u8(input_feature_dict["map_feature_valid"])
        features["tl_feature"] = input_feature_dict["tl_feature"]
        # (the masks of validation / training force every valid agent up to step_warm_start = time_step_current; a mask the
        # caller made itself is checked against that promise, and the warm start is not batched when it does not hold)
        w = self._warm_start_steps(teacher_forcing_cfg or self.hparams["teacher_forcing_reactive_replay"], batch["gt"])
        return self.rollout(
            features, latent=latent, goal=goal, goal_valid=goal_valid, mask_teacher_forcing=mask_teacher_forcing,
            deterministic_latent=deterministic_latent, deterministic_action=deterministic_action,
            step_start=self.hparams["time_step_sim_start"], step_end=self.hparams["time_step_end"], k_futures=1,
            gt=batch["gt"], latent_posterior=latent_is_posterior, require_vis_dict=require_vis_dict, rule_checker_tl=batch["gt"],
            latent_eps=latent_eps, warm_start_steps=w, action_eps=action_eps, generator=generator, hidden_drop=hidden_drop,
        )

    def finish_rollout(self) -> RolloutBuffer:
        """`RolloutBuffer.finish()` for a stepwise rollout (`buffer.py:72-90`): the buffer over the steps taken so far
        (slots of steps not yet simulated are undefined)."""
        out = self.engine._step_out
        if getattr(self, "_rule_ctx", None) is not None:
            features, k_futures, flags, tl, agent_goal = self._rule_ctx
            out.update(self.engine.rule_checks(features, out["check_state"], out["check_valid"], k_futures, flags, tl=tl,
                                               agent_goal=agent_goal))
        return self._buffer_from(out)

    def forward(self, map_feature: Optional[Tensor] = None, map_valid: Optional[Tensor] = None, tl_feature: Optional[Tensor] = None,
                tl_valid: Optional[Tensor] = None, goal_feature: Optional[Tensor] = None, goal_valid: Optional[Tensor] = None,
                action_override: Optional[Tensor] = None, mask_action_override: Optional[Tensor] = None,
                state_override: Optional[Dict[str, Tensor]] = None, mask_state_override: Optional[Tensor] = None,
                deterministic_action: Optional[bool] = None, require_train_dict: bool = True, require_vis_dict: bool = False,
                gt_valid: Optional[Tensor] = None, _skip_state: bool = False):
        """One simulation step with the reference's signature, the stateful `WaymoMotion.forward` (`waymo_motion.py:108-203`):
        advances the simulator opened by `rollout(..., stepwise=True)` and returns `(agent_state, agent_valid, train_dict, vis_dict)`.

        * `state_override` ({"agent_state" [N,A,4], "vel" [N,A,2], "acc" [N,A,1], "yaw_rate" [N,A,1]}, `Dynamics.state_keys`) and
          `mask_state_override` [N,A] are APPLIED to this step (`Dynamics.override_states`, `dynamics.py:132-149`; also how agents
          are spawned) -- this is how the reference's `rollout()` teacher-forces, `waymo_motion.py:269-306`.  A mask without a
          state dict must be all False (the reference passes `state_override=None` then).
        * With BOTH left at None the step takes the overrides bound when the rollout was opened (step t of the history arrays and of
          `mask_teacher_forcing`): the short form `forward()` of this mirror.  "No override at this step" is a zero mask.
        * The kill rule (`Dynamics.kill(violations, gt_valid)`, which the reference's loop calls right after `forward`, `:311-312`),
          the rule checks, the navigator and the buffer write are part of the same kernel launch, so `gt_valid` [N,A] (the ground
          truth's validity at this step, None = kill every agent that leaves the map) is an argument HERE, and the returned
          `agent_valid` is the validity after the kill.
        * The feature arguments are those bound at `rollout(stepwise=True)`: map / goal features are loop invariants whose K/V
          and fusion halves were hoisted there, the traffic-light step follows the step counter (`step_tl = min(step - 1, n - 1)`,
          `:291`), `goal_valid` is maintained by the navigator inside the kernel.  They are accepted for signature parity and
          checked for shape only.
        * `action_override` [N,A,2] (acceleration m/s^2, yaw rate rad/s) + `mask_action_override` [N,A] replace the policy's physical
          action of this step for the agents that are valid before it (`Dynamics.update`, `dynamics.py:96-100`); `action_log_prob`
          stays that of the policy's own action, as in the reference.
        * `require_vis_dict=True` (`:167,191-201`; the rollout must have been opened by `rollout(..., require_vis_dict=True)` or the loop
          driven through it): the fused step never materialises attention weights, so the step's `TrafficBots.forward` is ALSO run
          un-fused (`tb_forward`, need_weights) on the simulator's current inputs -- validity and GRU state of `tb_rollout_state`, the
          agent feature the fused launch of the previous step left in the every-step tap, the traffic lights of `step_tl`, the
          navigator's goal validity -- and `vis_dict` = {"action", "goal_valid", "attn_weights_to_pl / _tl / _agent"} on the host, as
          the reference returns it.  Visualisation path: ~70 small launches + 5 device-to-host copies per step.
        * Sampled actions (`deterministic_action=False`) work step by step as well: open the simulator with
          `rollout(..., deterministic_action=False, action_eps=..., stepwise=True)`; step s takes the draws `action_eps[:, :, s]`."""
        if (action_override is None) != (mask_action_override is None):
            raise ValueError("forward: action_override and mask_action_override go together")
        eng = self.engine
        if not getattr(eng, "_step_open", False):
            raise RuntimeError("forward: no stepwise rollout is open (call rollout(..., stepwise=True) first)")
        o = eng._step_out
        # `deterministic_action` (`:120`, `Dynamics.update(deterministic=)`): the draws of a sampled rollout are bound when the simulator is
        # opened (`rollout(deterministic_action=False, action_eps=...)`, one [2]-vector per instance, agent and step), so a call can only
        # confirm the mode, not switch it; None = as opened
        if deterministic_action is not None and bool(deterministic_action) == bool(o.get("_sampled_actions", False)):
            raise ValueError("forward: deterministic_action must match the mode the simulator was opened with "
                             "(rollout(..., deterministic_action=False, action_eps=...) binds the draws of a sampled rollout)")
        n, a = o["preds"].shape[:2]
        for name, ten, shape in (("map_feature", map_feature, (None, None, 128)), ("tl_feature", tl_feature, (None, None, 128)),
                                 ("goal_feature", goal_feature, (n, a, 128)), ("goal_valid", goal_valid, (n, a))):
            if ten is not None and (ten.dim() != len(shape) or any(e is not None and e != g for e, g in zip(shape, ten.shape))):
                raise ValueError(f"forward: {name} has shape {tuple(ten.shape)}, expected {shape}")
        override = None
        if state_override is not None or mask_state_override is not None:
            if mask_state_override is None:
                raise ValueError("forward: state_override needs mask_state_override")
            mask = mask_state_override
            if not isinstance(mask, Tensor):
                raise ValueError("forward: mask_state_override must be a tensor")
            if state_override is None:
                # the reference passes state_override=None only with an all-False mask (waymo_motion.py:274-276) and would fail on
                # `state_override[...]` otherwise: a host mask is checked, a device mask is forced to all-False instead of syncing
                if mask.device.type == "cpu" and bool(mask.any()):
                    raise ValueError("forward: mask_state_override is set somewhere but state_override is None")
                z = torch.zeros(n, a, 4, device=self.device)
                state_override = {"agent_state": z, "vel": z[..., :2], "acc": z[..., :1], "yaw_rate": z[..., :1]}
                mask = torch.zeros(n, a, dtype=torch.uint8, device=self.device)
            override = {"mask": mask, "agent_state": state_override["agent_state"], "vel": state_override["vel"],
                        "acc": state_override["acc"], "yaw_rate": state_override["yaw_rate"], "gt_valid": gt_valid}
        elif gt_valid is not None:
            raise ValueError("forward: gt_valid goes with state_override / mask_state_override (the bound history supplies its own)")
        if action_override is not None:
            override = dict(override or {}, action=action_override, action_mask=mask_action_override)
        vis = getattr(self, "_vis", None)
        fw = None
        if require_vis_dict:
            if vis is None or "tap_agent_feature" not in o:
                raise RuntimeError("forward(require_vis_dict=True): open the simulator with rollout(..., require_vis_dict=True)")
            st0 = eng.rollout_state()
            step_tl = min(self._step_t - 1, vis["tl_valid"].shape[1] - 1)  # `waymo_motion.py:291`
            fw = eng.forward_trunk(st0["agent_valid"].bool(), o["tap_agent_feature"], vis["map_valid"], vis["map_feature"],
                                   vis["rep"](vis["tl_valid"][:, step_tl]), vis["rep"](vis["tl_feature"][:, step_tl]), vis["goal_valid"],
                                   vis["goal_feature"], vis["z"], st0["hidden"].flatten(1, 2), need_weights=True)
        eng.rollout_step(override)
        s_idx = self._step_t - self.hparams["time_step_sim_start"]
        self._step_t += 1
        vis_dict = {}
        if require_vis_dict:
            vis_dict = {"action": o["actions"][:, :, s_idx].cpu(), "goal_valid": vis["goal_valid"].cpu(),
                        "attn_weights_to_pl": fw["attn_pl"].cpu(), "attn_weights_to_tl": fw["attn_tl"].cpu(),
                        "attn_weights_to_agent": fw["attn_agent"].cpu()}
            # the navigator (`GoalManager.disable_goal_reached`, goal_manager.py:155-162) for the next step: goals of agents that are
            # gone or have reached their destination are switched off (flag logic on recorded outputs, no arithmetic)
            st1 = eng.rollout_state()
            vis["goal_valid"] = vis["goal_valid"] & st1["agent_valid"].bool() & ~o["dest_reached"][:, :, s_idx].bool()
            if _skip_state:
                return None, None, {}, vis_dict
        if _skip_state:  # (the per-step loops of this mirror read the state once at the end: no three D2D copies per step)
            return None, None, {}, {}
        st = eng.rollout_state()
        train_dict = {}
        if require_train_dict:
            train_dict = {
                "latent_log_prob": o["latent_log_prob"], "action_log_prob": o["action_log_probs"][:, :, s_idx],
                "pred_valid": o["valid"][:, :, s_idx].bool(), "pred_state": o["preds"][:, :, s_idx],
            }
        self.model.hidden = st["hidden"].flatten(1, 2)
        return st["agent_state"], st["agent_valid"].bool(), train_dict, vis_dict

    def _rollout_with_per_step_overrides(self, features: Dict[str, Tensor], mask_teacher_forcing: Tensor, k_futures: int,
                                         step_start: int, step_end: int, gt: Optional[Dict[str, Tensor]],
                                         gt_sdc: Optional[Dict[str, Tensor]], require_vis_dict: bool = False) -> None:
        """The reference's loop (`waymo_motion.py:269-306`) over `forward(state_override=..., mask_state_override=...)` for a
        simulator already opened with `stepwise=True`: per step the override of step t of the teacher-forcing source, plus -- the
        what-if motion prediction of `gt_sdc` ({"agent_state" [N,S,4], "vel" [N,S,2], "acc" / "yaw_rate" [N,S,1]}) -- agent 0 forced
        to the given trajectory at EVERY step (`:279-284`)."""
        src = gt if gt is not None else features
        n_src = src["agent_valid"].shape[1]
        rep = (lambda x: x.repeat_interleave(k_futures, 0)) if k_futures > 1 else (lambda x: x)
        mask_tf = mask_teacher_forcing.to(self.device).bool()
        for t in range(step_start, step_end + 1):
            in_src = t < n_src
            m = rep(mask_tf[:, t]).clone() if in_src else torch.zeros_like(rep(mask_tf[:, 0]))
            tt = t if in_src else 0
            so = {"agent_state": rep(src["agent_state"][:, tt]).clone(), "vel": rep(src["agent_vel"][:, tt]).clone(),
                  "acc": rep(src["agent_acc"][:, tt]).unsqueeze(-1).clone(), "yaw_rate": rep(src["agent_yaw_rate"][:, tt]).unsqueeze(-1).clone()}
            if gt_sdc is not None:
                m[:, 0] = True
                for k in gt_sdc.keys():  # (the reference iterates gt_sdc.keys(), waymo_motion.py:283-284: a dict with fewer keys is fine)
                    so[k][:, 0] = gt_sdc[k][:, t].to(self.device).to(so[k].dtype).reshape(so[k][:, 0].shape)
            gtv = rep(src["agent_valid"][:, t]) if in_src else None
            vd = self.forward(state_override=so, mask_state_override=m, gt_valid=gtv, require_train_dict=False, _skip_state=True,
                              require_vis_dict=require_vis_dict)[3]
            if require_vis_dict:
                self._vis["collected"].append(vd)

    def joint_future_pred(
        self,
        batch: Dict[str, Tensor],
        input_feature_dict: Dict[str, Tensor],
        latent: DiagGaussian,
        goal: DestCategorical,
        goal_valid: Tensor,
        require_vis_dict: bool = False,
        latent_eps: Optional[Tensor] = None,
        goal_sample: Optional[Tensor] = None,
        generator=None,
        tap_step: int = -1,
        action_eps: Optional[Tensor] = None,
    ) -> Tuple[RolloutBuffer, Tensor, Tensor]:
        """`WaymoMotion.joint_future_pred` (`waymo_motion.py:478-572`): K futures per scene, sample 0
        deterministic.  `batch` is the pre-processed scene.  The reference materialises every tensor K times
        with repeat_interleave; here only the per-instance tensors are."""
        k = self.hparams["n_joint_future"]
        b, _, a = batch["agent_valid"].shape
        deterministic = self._det_cache.get((b, k, a))  # (a constant of the shape: sample 0 of every scene takes the mean)
        if deterministic is None:
            deterministic = torch.zeros(b * k, a, dtype=torch.bool, device=self.device)
            deterministic[::k] = True
            self._det_cache[(b, k, a)] = deterministic
        latent.repeat_interleave_(k, 0)
        goal.repeat_interleave_(k, 0)
        if goal_sample is None:
            goal_sample = goal.sample(deterministic, generator=generator)
        else:
            if not goal_sample.is_cuda:  # a caller-supplied destination (host tensor: checked for free; the device clamps anyway)
                n_pl = input_feature_dict["map_feature"].shape[1]
                if int(goal_sample.min()) < 0 or int(goal_sample.max()) >= n_pl:
                    raise IndexError(f"goal_sample holds a polyline index outside [0, {n_pl})")
            goal_sample = goal_sample.to(self.device).reshape(b * k, a)
        goal_log_probs = goal.log_prob(goal_sample)
        if k > 1:
            goal_valid = goal_valid.repeat_interleave(k, 0)
        features = {k_: v for k_, v in batch.items() if k_ != "gt"}
        features["map_feature"] = input_feature_dict["map_feature"]
        features["map_feature_valid"] = as_u8(input_feature_dict["map_feature_valid"])
        features["tl_feature"] = input_feature_dict["tl_feature"]
        tf = self.hparams["teacher_forcing_joint_future_pred"]
        # validation: features["agent_valid"] etc. are the 91-step ground truth (waymo_motion.py:538-545 with batch["agent/*"]
        # left untouched), which only matters for the kill rule; test_step overwrites them with the history (:925-926)
        gt = batch.get("gt")
        src = gt if gt is not None else batch
        if src.get("_tf_params") == (tf.get("step_spawn_agent", 10), tf.get("step_warm_start", 10)):
            mask_tf = src["_tf_mask"]  # made on the host while the batch was staged (staging.teacher_forcing_mask_np)
        else:
            mask_tf = teacher_forcing_mask(as_bool(src["agent_valid"]), tf.get("step_spawn_agent", 10), tf.get("step_warm_start", 10))
        buf = self.rollout(
            features, latent=latent, goal=goal_sample, goal_valid=goal_valid, mask_teacher_forcing=mask_tf,
            deterministic_latent=deterministic, deterministic_action=action_eps is None,  # (the reference passes True, :560)
            action_eps=action_eps,
            step_start=self.hparams["time_step_sim_start"], step_end=self.hparams["time_step_end"],
            k_futures=k, latent_eps=latent_eps, tap_step=tap_step, gt=gt, generator=generator,  # (round 5: the personalities of the
            # sampled futures were drawn from torch's GLOBAL generator whatever `generator` said: not repeatable per seed)
            warm_start_steps=self._warm_start_steps(tf, gt if gt is not None else batch), require_vis_dict=require_vis_dict,
        )
        buf.flatten_repeat(k)
        goal_log_probs = goal_log_probs.view(b, k, a).transpose(1, 2)
        goal_sample = goal_sample.view(b, k, a).transpose(1, 2)
        return buf, goal_sample, goal_log_probs

    @_range_fallback
    def test_step(self, batch: Dict[str, Tensor], batch_idx: int = 0, latent_eps=None, goal_sample=None, generator=None,
                  tap_step: int = -1, action_eps: Optional[Tensor] = None) -> Dict[str, Tensor]:
        """`WaymoMotion.test_step` (`waymo_motion.py:902-940`) up to and including `waymo_post_processing`; the submission writer
        (`:942-949`) is out of scope.  Returns the buffer, the intermediate products and the post-processed `pred_dict`.
        `action_eps` (extension, [B*K, A, S, 2]): run the futures with SAMPLED actions instead of the reference's deterministic ones."""
        scene = self.pre_processing(batch)
        scene.pop("gt", None)  # batch["agent/*"] = batch["history/agent/*"] (waymo_motion.py:925-926) ...
        if "hist" in scene:    # ... over ALL history steps of the batch, also beyond time_step_current (runtime.hist_from_batch)
            scene["gt"] = scene.pop("hist")
        input_feature_dict = self.model.encode_input_features(scene)
        goal_valid = as_bool(scene["_goal_valid"]) if "_goal_valid" in scene else as_bool(scene["agent_valid"]).any(1)
        goal_pred = self.model.goal_manager.pred_goal()
        latent_prior = self.model.latent_encoder()
        latent_mean, latent_valid = latent_prior.mean, latent_prior.valid  # (repeat_interleave_ below rebinds them)
        buf, gs, glp = self.joint_future_pred(
            scene, input_feature_dict, latent_prior, goal_pred, goal_valid, latent_eps=latent_eps, goal_sample=goal_sample,
            generator=generator, tap_step=tap_step, action_eps=action_eps,
        )
        scores = torch.exp(buf.latent_log_probs[..., 0] + glp)  # waymo_motion.py:936
        pred_dict = None
        if buf.preds.shape[3] > buf.step_future_start:  # (a rollout that stops at the current step has no future to post-process)
            pred_dict = self.waymo_post_processing(
                valid=buf.valid[:, :, 0].any(-1), scores=scores, trajs=buf.preds[:, :, :, buf.step_future_start:],
                agent_type=scene["agent_type"],
            )
        self._after_enqueue(batch)
        return {
            "rollout_buffer": buf, "goal_sample": gs, "goal_log_probs": glp, "input_feature_dict": input_feature_dict,
            "latent_mean": latent_mean, "latent_valid": latent_valid, "dest_logits": self.model._enc["dest_logits"],
            "scores": scores, "pred_dict": pred_dict,
        }

    @staticmethod
    def _after_enqueue(batch) -> None:
        """Called by the harness steps once their GPU work is enqueued and before they synchronise: a batch that came out of
        `prefetch` lets its prefetcher stage (and encode) the NEXT batch now, under this batch's rollout."""
        pf = getattr(batch, "prefetcher", None)
        if pf is not None:
            pf.advance()

    def _check_range(self) -> None:
        """`tb_check_status` (one stream synchronisation; `self.check_range = False` skips it): raises if an fp16-pair operand of the
        fp32-accurate kernels left the fp16 range.  The three harness steps do more than raise: see `_range_fallback`."""
        if self.check_range:
            self.engine.check_status()

    def _metric_holders(self):
        return [getattr(self, n) for n in ("train_metrics_reactive_replay", "err_metrics_reactive_replay", "rule_metrics_reactive_replay",
                                           "err_metrics_joint_future_pred", "rule_metrics_joint_future_pred") if hasattr(self, n)]

    @_range_fallback
    def training_step(self, batch: Dict[str, Tensor], batch_idx: int = 0, latent_eps: Optional[Tensor] = None,
                      rollout_prior: bool = False, current_epoch: int = 0, action_eps: Optional[Tensor] = None,
                      generator=None, irrelevant_draw: Optional[Tensor] = None, history_keep: Optional[Dict[str, Tensor]] = None,
                      hidden_drop=None, latent_perturb: Optional[Dict[str, Tensor]] = None) -> Dict[str, object]:
        """`latent_perturb` = {"yaw": [B], "pos": [B,2]} uniform draws in [0, 1) for `pre_processing.latent.perturb_input_to_latent`
        (:meth:`_perturb_latent_inputs`; drawn here with `generator` when the config asks for it and none are given).
        Train-mode Bernoulli masks (explicit draws, the `irrelevant_draw` pattern; drawn here with `generator` when the config asks
        for them and none are given):
          * `history_keep` -- KEEP masks of `pre_processing.input.dropout_p_history` ("input_agent" [B,10,A]: the agent history but its
            last step, "input_tl" [B,11,T], "input_map" [B,P,20]; `sc_input.py:100-106`) and of `pre_processing.latent.dropout_p_history`
            ("post_tl" [B,91,T], "post_agent" [B,91,A]: the posterior's inputs, `sc_latent.py:171-173,216-218`).  As in the reference the
            posterior's masks also apply to the ground-truth validity the replay runs against, and the input's map mask to the rule
            checker's map (both are in-place edits of aliased tensors there);
          * `hidden_drop` [S] -- `p_drop_hidden`: the GRU state of the whole batch is zeroed after the steps where it is set
            (`waymo_motion.py:345-351`).
        nn.Dropout inside the network stays off (eval-mode arithmetic).
        Forward value of `WaymoMotion.training_step` (`waymo_motion.py:356-418`): the episode replayed under
        `teacher_forcing_training` with a SAMPLE of the posterior personality (or of the prior when `rollout_prior`, which the
        reference decides with `torch.rand(1) < p_training_rollout_prior`) and the ground-truth destination, then
        `TrainingMetrics` -> {"training/loss", "training/vae_kl", "training/diffbar_reward", "training/goal_loss"}.
        Random draws are explicit (`latent_eps` [B,A,16] standard normal; None = the mean), the network runs without dropout
        (the reference's eval-mode arithmetic), and there is no backward pass: this is the loss a validation of the training
        objective reports, not an optimisation step."""
        hp = self.hparams
        det_action = bool(hp.get("training_deterministic_action", True))  # False: sampled actions (`action_eps` or torch.randn)
        tf = hp["teacher_forcing_training"]
        if tf.get("step_horizon", 0) - tf.get("step_horizon_decrease_per_epoch", 0) * current_epoch > 0 or \
                tf.get("prob_forcing_agent", 0) - tf.get("prob_forcing_agent_decrease_per_epoch", 0) * current_epoch > 0:
            raise NotImplementedError("teacher_forcing_training schedules (step_horizon / prob_forcing_agent) are not built")
        scene = self.pre_processing(batch)
        scene.pop("hist", None)  # (test_step only: validation / training roll out on the ground truth, `waymo_motion.py:538-545`)
        if "gt" not in scene:
            raise ValueError("training_step needs a training / validation batch (agent/*, tl_stop/* ground truth)")
        gt = scene["gt"]
        scene, gt, hidden_drop = self._train_mode_masks(scene, gt, history_keep, hidden_drop, generator)
        input_feature_dict = self.model.encode_input_features(scene)
        goal_gt, goal_valid = self.model.goal_manager.get_gt_goal(scene["agent_valid"], gt.get("gt_goal"), gt["gt_dest"])
        goal_pred = self.model.goal_manager.pred_goal()
        if hp["pre_processing"].get("latent", {}).get("perturb_input_to_latent", False):
            # both personality encoders see the episode in a frame drawn per scene (their own encode of the re-centred inputs; the
            # policy's features, the destination predictor and the replay stay in the scene frame)
            b_ = scene["agent_valid"].shape[0]
            lp = latent_perturb if latent_perturb is not None else {
                "yaw": torch.rand(b_, device=self.device, generator=generator), "pos": torch.rand(b_, 2, device=self.device, generator=generator)}
            # The re-centred inputs carry the INPUT-DROPPED validity: `SceneCentricInput` aliases input/agent_valid, input/tl_valid and
            # input/map_valid to the sc/* tensors and masks them IN PLACE (`sc_input.py:103-113`: `batch["input/agent_valid"] =
            # batch["sc/agent_valid"]` ... `&=`), so the sc/* validity `SceneCentricLatent` reads under perturbation
            # (`sc_latent.py:142-167`) is the masked one -- and sc/map_valid IS batch["map/valid"], which the posterior's map shares.
            # The posterior's agents / traffic lights come from agent/valid, tl_stop/valid (separate tensors: not masked by the input
            # dropout).  Pinned by golden `train_perturb_dropout`, made by the reference (round 6; the advisor's reading of round 5 --
            # "from the un-dropped sc/*" -- overlooked the aliasing, and the change it prompted is undone here).
            scene_l, gt_l = self._perturb_latent_inputs(scene, gt, lp["yaw"], lp["pos"])
            enc_l = self.engine.encode_scene(scene_l)
            post = self.engine.encode_posterior(gt_l, enc_l)
            latent_post = DiagGaussian(post["latent_mean"], self.model._log_std_post, valid=as_bool(post["latent_valid"]), engine=self.engine)
            latent_prior = DiagGaussian(enc_l["latent_mean"], self.model._log_std, valid=as_bool(enc_l["latent_valid"]), engine=self.engine)
        else:
            latent_post = self.model.latent_encoder(posterior=True, gt=gt)
            latent_prior = self.model.latent_encoder()
        latent = latent_prior if rollout_prior else latent_post
        mask_tf = teacher_forcing_mask(gt["agent_valid"].bool(), tf.get("step_spawn_agent", 10), tf.get("step_warm_start", 10))
        buf = self.reactive_replay(scene, input_feature_dict, mask_tf, latent, goal_gt, goal_valid,
                                   deterministic_latent=latent_eps is None, deterministic_action=det_action, latent_eps=latent_eps,
                                   latent_is_posterior=not rollout_prior, teacher_forcing_cfg=tf, action_eps=action_eps,
                                   generator=generator, hidden_drop=hidden_drop)
        gv, gs = self._gt_slices(gt, hp["time_step_sim_start"], hp["time_step_end"])
        raw = {"valid": buf.valid, "preds": buf.preds, "override_masks": buf.override_masks}
        _, _, states = self.engine.train_partials(
            raw, gv, gs, scene["agent_size"], dest_logits=self.model._enc["dest_logits"], goal_valid=goal_pred.valid, gt_dest=goal_gt,
            post={"latent_mean": latent_post.mean, "latent_valid": latent_post.valid},
            prior={"latent_mean": latent_prior.mean, "latent_valid": latent_prior.valid},
            agent_role=gt["agent_role"], irrelevant_draw=irrelevant_draw, generator=generator)
        m = TrainingMetrics("training", **hp["training_metrics"])
        m.update(states)
        out = m.compute()  # (the reference logs and resets per step, :415-417)
        return {"loss": out["training/loss"], "metrics_dict": out, "train_states": states, "rollout_buffer": buf,
                "latent_post": latent_post, "latent_prior": latent_prior}

    def _perturb_latent_inputs(self, scene: Dict[str, Tensor], gt: Dict[str, Tensor], yaw_u: Tensor, pos_u: Tensor):
        """`pre_processing.latent.perturb_input_to_latent` (train mode; `sc_latent.py:115-124` and the `torch_*2local` calls below it):
        the inputs of BOTH personality encoders -- map, traffic lights, agents; history and 91-step ground truth -- are moved into a
        frame drawn per scene, yaw = u * 2 max_rad - max_rad, position = u * 2 max_meter - max_meter (`yaw_u` [B], `pos_u` [B,2]:
        the uniform draws, explicit as everywhere in this mirror): p' = (p - t) R, d' = d R, yaw' = yaw - yaw_0 (not wrapped),
        R = [[cos, -sin], [sin, cos]] (`transform_utils.py:121-131,146-157,174-184,200-213`).  Returns copies of (scene, gt)."""
        lat = self.hparams["pre_processing"].get("latent", {})
        max_meter, max_rad = float(lat.get("max_meter", 50.0)), float(lat.get("max_rad", 3.14))
        dev, f32 = self.device, torch.float32
        yaw = yaw_u.to(dev, f32) * 2 * max_rad - max_rad          # [B]
        pos = pos_u.to(dev, f32) * 2 * max_meter - max_meter      # [B,2]
        c, s_ = torch.cos(yaw), torch.sin(yaw)
        rot = torch.stack([torch.stack([c, -s_], -1), torch.stack([s_, c], -1)], -2)  # [B,2,2]

        def pos2local(x):   # [B, ..., 2]
            sh = x.shape
            return torch.matmul(x.reshape(sh[0], -1, 2) - pos[:, None, :], rot).reshape(sh).contiguous()

        def dir2local(x):
            sh = x.shape
            return torch.matmul(x.reshape(sh[0], -1, 2), rot).reshape(sh).contiguous()

        def agents(d):
            o = dict(d)
            o["agent_pos"] = pos2local(d["agent_pos"])
            o["agent_vel"] = dir2local(d["agent_vel"])
            o["agent_yaw"] = (d["agent_yaw"] - yaw.reshape(-1, *([1] * (d["agent_yaw"].dim() - 1)))).contiguous()
            o["agent_state"] = torch.cat([o["agent_pos"], o["agent_yaw"].unsqueeze(-1), d["agent_spd"].unsqueeze(-1)], -1).contiguous()
            o["tl_pos"] = pos2local(d["tl_pos"])
            o["tl_dir"] = dir2local(d["tl_dir"])
            return o

        scene_l = agents(scene)
        scene_l["map_pos"] = pos2local(scene["map_pos"])
        scene_l["map_dir"] = dir2local(scene["map_dir"])
        scene_l.pop("gt", None)
        return scene_l, agents(gt)

    def _train_mode_masks(self, scene: Dict[str, Tensor], gt: Dict[str, Tensor], history_keep, hidden_drop, generator):
        """Applies / draws the train-mode Bernoulli masks of :meth:`training_step`; returns (scene, gt, hidden_drop) -- copies of the
        dicts with the masked validity tensors (the caller's batch is not edited)."""
        hp = self.hparams
        p_in = float(hp["pre_processing"]["input"].get("dropout_p_history", -1))
        p_lat = float(hp["pre_processing"].get("latent", {}).get("dropout_p_history", -1))
        p_hid = float(hp.get("p_drop_hidden", -1.0))
        if hp["pre_processing"].get("latent", {}).get("perturb_input_to_latent", False) and 0 < p_lat <= 1.0:
            raise NotImplementedError("pre_processing.latent.perturb_input_to_latent together with pre_processing.latent.dropout_p_history: "
                                      "the prior-side history masks of the re-centred inputs (sc_latent.py:158-160,190-192) are not built")
        dev = self.device
        keep = dict(history_keep) if history_keep is not None else {}

        def draw(name, shape, p):
            if name not in keep:
                keep[name] = torch.bernoulli(torch.full(shape, 1.0 - p, device=dev), generator=generator).bool()
            return keep[name].to(dev).bool().reshape(shape)

        b, nh, a = scene["agent_valid"].shape
        if 0 < p_in <= 1.0 or any(k in keep for k in ("input_agent", "input_tl", "input_map")):
            av = scene["agent_valid"].bool().clone()
            av[:, :-1] &= draw("input_agent", (b, nh - 1, a), p_in)
            tlv = scene["tl_valid"].bool() & draw("input_tl", tuple(scene["tl_valid"].shape), p_in)
            mv = scene["map_valid"].bool() & draw("input_map", tuple(scene["map_valid"].shape), p_in)
            scene = dict(scene, agent_valid=av.to(torch.uint8), tl_valid=tlv.to(torch.uint8).contiguous(), map_valid=mv.to(torch.uint8).contiguous(),
                         warm_ok=False)
            for stale in ("_goal_valid", "_tf_mask", "_tf_params"):  # (host-made from the unmasked validity when the batch was staged)
                scene.pop(stale, None)
            scene.update({"input/agent_valid": av, "input/tl_valid": tlv, "input/map_valid": mv,
                          "latent_prior/agent_valid": av, "latent_prior/tl_valid": tlv, "latent_prior/map_valid": mv})
        if 0 < p_lat <= 1.0 or any(k in keep for k in ("post_agent", "post_tl")):
            gav = gt["agent_valid"].bool() & draw("post_agent", tuple(gt["agent_valid"].shape), p_lat)
            gtl = gt["tl_valid"].bool() & draw("post_tl", tuple(gt["tl_valid"].shape), p_lat)
            gt = dict(gt, agent_valid=gav.to(torch.uint8).contiguous(), tl_valid=gtl.to(torch.uint8).contiguous(), warm_ok=False)
            gt.pop("_tf_mask", None)
            gt.pop("_tf_params", None)
            scene = dict(scene, gt=gt)
        if b == 1 and (scene.get("warm_ok") is False or gt.get("warm_ok") is False):
            # A batch of ONE scene: in the reference `sc/agent_valid = batch["agent/valid"][:, :n_step_hist].contiguous()` (likewise the
            # traffic lights; `scene_centric.py:92-99`) is a VIEW when the batch dimension is 1 -- the slice is contiguous already -- so
            # the history and the first steps of the ground truth share storage, and every in-place `&=` of the two pre-processing
            # modules lands on both (found by tools/fuzz_oracle_vs_reference.py --train at n_scene = 1; with >= 2 scenes it is a copy)
            av_all = scene["agent_valid"].bool() & gt["agent_valid"][:, :nh].bool()
            tl_all = scene["tl_valid"].bool() & gt["tl_valid"][:, :nh].bool()
            ga, gtl = gt["agent_valid"].clone(), gt["tl_valid"].clone()
            ga[:, :nh], gtl[:, :nh] = av_all.to(ga.dtype), tl_all.to(gtl.dtype)
            gt = dict(gt, agent_valid=ga.contiguous(), tl_valid=gtl.contiguous(), warm_ok=False)
            gt.pop("_tf_mask", None)
            gt.pop("_tf_params", None)
            scene = dict(scene, agent_valid=av_all.to(torch.uint8).contiguous(), tl_valid=tl_all.to(torch.uint8).contiguous(), warm_ok=False, gt=gt)
            for stale in ("_goal_valid", "_tf_mask", "_tf_params"):
                scene.pop(stale, None)
            scene.update({"input/agent_valid": av_all, "input/tl_valid": tl_all, "latent_prior/agent_valid": av_all, "latent_prior/tl_valid": tl_all})
        if hidden_drop is None and p_hid > 0:
            n_step = hp["time_step_end"] - hp["time_step_sim_start"] + 1
            hidden_drop = (torch.rand(n_step, device=dev, generator=generator) < p_hid).cpu()
        return scene, gt, hidden_drop

    @_range_fallback
    def validation_step(self, batch: Dict[str, Tensor], batch_idx: int = 0, latent_eps=None, goal_sample=None, generator=None,
                        irrelevant_draw: Optional[Tensor] = None) -> Dict[str, object]:
        """`WaymoMotion.validation_step` (`waymo_motion.py:574-735`) without the WOMD-metric ops, submission writers and videos:
        posterior / prior personalities and the destination prediction, `reactive_replay` (posterior mean, ground-truth
        destination, teacher_forcing_reactive_replay) with its error / traffic-rule / training metric states and post-processed
        prediction, then `joint_future_pred` (prior samples, predicted destinations) with its error / traffic-rule states.
        The metric holders accumulate over calls; `validation_epoch_end`-style reporting = `.sync()` + `.compute()` on them."""
        scene = self.pre_processing(batch)
        scene.pop("hist", None)  # (test_step only: validation / training roll out on the ground truth, `waymo_motion.py:538-545`)
        if "gt" not in scene:
            raise ValueError("validation_step needs a validation batch (agent/*, tl_stop/* ground truth)")
        gt = scene["gt"]
        hp = self.hparams
        s0, s1 = hp["time_step_sim_start"], hp["time_step_end"]
        if gt["agent_valid"].shape[1] <= s1:
            raise ValueError("ground truth shorter than time_step_end")
        input_feature_dict = self.model.encode_input_features(scene)
        goal_gt, goal_valid = self.model.goal_manager.get_gt_goal(scene["agent_valid"], gt.get("gt_goal"), gt["gt_dest"])
        goal_pred = self.model.goal_manager.pred_goal()
        latent_post = self.model.latent_encoder(posterior=True, gt=gt)
        latent_prior = self.model.latent_encoder()
        prior_mean, prior_valid = latent_prior.mean, latent_prior.valid
        # ---- reactive replay
        tf = hp["teacher_forcing_reactive_replay"]
        if gt.get("_tf_params") == (tf.get("step_spawn_agent", 10), tf.get("step_warm_start", 10)):
            mask_tf = gt["_tf_mask"]  # (the same parameters as joint_future_pred's: the mask the staging made on the host)
        else:
            mask_tf = teacher_forcing_mask(as_bool(gt["agent_valid"]), tf.get("step_spawn_agent", 10), tf.get("step_warm_start", 10))
        buf = self.reactive_replay(scene, input_feature_dict, mask_tf, latent_post, goal_gt, goal_valid,
                                   deterministic_latent=True, deterministic_action=True)
        gv, gs = self._gt_slices(gt, s0, s1)
        raw = {"valid": buf.valid, "preds": buf.preds, "override_masks": buf.override_masks}
        _, _, train_states = self.engine.train_partials(
            raw, gv, gs, scene["agent_size"], dest_logits=self.model._enc["dest_logits"], goal_valid=goal_pred.valid, gt_dest=goal_gt,
            post={"latent_mean": latent_post.mean, "latent_valid": latent_post.valid},
            prior={"latent_mean": prior_mean, "latent_valid": prior_valid},
            agent_role=gt["agent_role"], irrelevant_draw=irrelevant_draw, generator=generator)
        buf.flatten_repeat(1)
        states = self.engine.metric_partials(buf.valid, buf.preds, buf.override_masks, buf.violations, scene["agent_type"],
                                             gt["agent_role"], gt_valid=gv, gt_states=gs, loss_for_teacher_forcing=False)
        self.err_metrics_reactive_replay.update(states[:4])
        self.rule_metrics_reactive_replay.update(states[4:])
        self.train_metrics_reactive_replay.update(train_states)
        # TrainingMetrics.update masks the first step_training_start steps of the buffer's `valid` IN PLACE when it takes no
        # copy first (training.py:87-93: p_loss_for_irrelevant <= 0 and loss_for_teacher_forcing), so the reference's
        # post-processing below sees that mask; mirrored for the same agents-with-a-prediction set
        tm = hp["training_metrics"]
        post_valid = buf.valid
        if tm["p_loss_for_irrelevant"] <= 0 and tm["loss_for_teacher_forcing"] and tm["step_training_start"] > 0:
            post_valid = buf.valid.clone()
            post_valid[..., : tm["step_training_start"]] = False
        pred_dict_rr = self.waymo_post_processing(
            valid=post_valid[:, :, 0].any(-1), scores=torch.ones_like(buf.preds[:, :, :, 0, 0]),
            trajs=buf.preds[:, :, :, buf.step_future_start:], agent_type=scene["agent_type"])
        # ---- joint future prediction
        buf_j, gsamp, glp = self.joint_future_pred(scene, input_feature_dict, latent_prior, goal_pred, goal_valid,
                                                   latent_eps=latent_eps, goal_sample=goal_sample, generator=generator)
        states_j = self.engine.metric_partials(buf_j.valid, buf_j.preds, buf_j.override_masks, buf_j.violations, scene["agent_type"],
                                               gt["agent_role"], gt_valid=gv, gt_states=gs, loss_for_teacher_forcing=False)
        self.err_metrics_joint_future_pred.update(states_j[:4])
        self.rule_metrics_joint_future_pred.update(states_j[4:])
        scores = torch.exp(buf_j.latent_log_probs[..., 0] + glp)
        pred_dict_j = self.waymo_post_processing(
            valid=buf_j.valid[:, :, 0].any(-1), scores=scores, trajs=buf_j.preds[:, :, :, buf_j.step_future_start:],
            agent_type=scene["agent_type"])
        self._after_enqueue(batch)
        return {
            "reactive_replay": {"rollout_buffer": buf, "train_states": train_states, "metric_states": states, "pred_dict": pred_dict_rr},
            "joint_future_pred": {"rollout_buffer": buf_j, "goal_sample": gsamp, "goal_log_probs": glp, "metric_states": states_j,
                                  "scores": scores, "pred_dict": pred_dict_j},
            "latent_post": latent_post, "latent_prior_mean": prior_mean, "latent_prior_valid": prior_valid,
            "dest_logits": self.model._enc["dest_logits"], "input_feature_dict": input_feature_dict,
        }
