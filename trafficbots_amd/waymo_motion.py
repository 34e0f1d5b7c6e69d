"""Drop-in surface of the reference's task module for the hot path.

Same method names, argument meaning and outputs as `pl_modules.waymo_motion.WaymoMotion`
(`src/pl_modules/waymo_motion.py`) and `models.traffic_bots.TrafficBots`
(`src/models/traffic_bots.py`) for: `pre_processing`, `encode_input_features`,
`goal_manager.pred_goal`, `latent_encoder`, `rollout`, `joint_future_pred`, `test_step`.
Everything numerical is executed by `libtrafficbots_hip.so` through :class:`HipEngine`; this file
is argument plumbing.  Lightning/Hydra are not required (and not rebuilt): the class is a plain
object that can be wrapped by a LightningModule in the reference harness.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple, Union

import torch
from torch import Tensor

from .config import load_model_config
from .distributions import DestCategorical, DiagGaussian
from .runtime import HipEngine, scene_from_batch, teacher_forcing_mask

_VIOLATION_KEYS = (
    "outside_map", "outside_map_this_step", "collided", "collided_this_step", "run_road_edge",
    "run_road_edge_this_step", "run_red_light", "run_red_light_this_step", "passive", "passive_this_step",
    "goal_reached", "goal_reached_this_step", "dest_reached", "dest_reached_this_step",
)


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
        self.diffbar_rewards = []  # training-only in the reference; not produced (OUT OF SCOPE)
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


class _GoalManager:
    goal_attr_mode = "dest"
    dummy = False
    update_goal = False

    def __init__(self, owner: "TrafficBots") -> None:
        self._o = owner

    def pred_goal(self, **kwargs) -> DestCategorical:
        """`GoalManager.pred_goal` -> `DestPredictor.forward` (`goal_manager.py:78-82,202-333`).  The logits
        were produced together with the features by `encode_input_features`."""
        return DestCategorical(logits=self._o._enc["dest_logits"], valid=self._o._enc["latent_valid"].bool())


class TrafficBots:
    """Facade over the HIP encoders with the reference's method names (`traffic_bots.py:109-161`)."""

    def __init__(self, engine: HipEngine) -> None:
        self.engine = engine
        self.goal_manager = _GoalManager(self)
        self._enc: Dict[str, Tensor] = {}
        self._log_std: Optional[Tensor] = None

    def encode_input_features(self, scene: Dict[str, Tensor]) -> Dict[str, Tensor]:
        """`TrafficBots.encode_input_features` (`traffic_bots.py:109-151`) on a pre-processed scene
        (:meth:`WaymoMotion.pre_processing`).  One HIP call also yields the prior mean and the destination
        logits, which `latent_encoder` / `goal_manager.pred_goal` then hand out."""
        self._enc = self.engine.encode_scene(scene)
        e = self._enc
        return {
            "agent_feature": e["agent_feature"], "agent_feature_valid": scene["agent_valid"].bool(),
            "map_feature": e["map_feature"], "map_feature_valid": e["map_feature_valid"].bool(),
            "tl_feature": e["tl_feature"], "tl_feature_valid": scene["tl_valid"].bool(),
        }

    def latent_encoder(self, **kwargs) -> DiagGaussian:
        """`LatentEncoder.forward`, prior branch (`latent_encoder.py:70-147`)."""
        return DiagGaussian(self._enc["latent_mean"], self._log_std, valid=self._enc["latent_valid"].bool())


class WaymoMotion:
    def __init__(self, config_path: Optional[str] = None, device: str = "cuda:0", **overrides) -> None:
        self.hparams = load_model_config(config_path, overrides or None)
        self.device = torch.device(device)
        self.engine = HipEngine(self.hparams, device)
        self.model = TrafficBots(self.engine)
        self.n_hist = self.hparams["time_step_current"] + 1
        from .post_processing import WaymoPostProcessing

        self.waymo_post_processing = WaymoPostProcessing(self.engine, **self.hparams.get("waymo_post_processing", {}))

    # ------------------------------------------------------------------ weights
    def load_state_dict(self, state_dict: Dict) -> None:
        self.engine.load_state_dict(state_dict)
        ls = state_dict["model.latent_encoder.latent_prior_dist.log_std"]
        self.model._log_std = torch.as_tensor(ls, dtype=torch.float32).to(self.device)

    # ------------------------------------------------------------------ pre-processing
    def pre_processing(self, batch: Dict[str, Tensor]) -> Dict[str, Tensor]:
        """Eval-mode `SceneCentricPreProcessing` + the layout half of `SceneCentricInput`
        (`scene_centric.py:103-133`, `sc_input.py:100-140`); attr/PE/MLP run inside `tb_encode_scene`."""
        return scene_from_batch(batch, self.device, self.n_hist)

    # ------------------------------------------------------------------ rollout
    def rollout(
        self,
        features: Dict[str, Tensor],
        latent: DiagGaussian,
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
    ) -> RolloutBuffer:
        """`WaymoMotion.rollout` (`waymo_motion.py:205-354`).  `stepwise=True` only initialises the simulator
        (`model.init`, `dynamics.init`, goal features: `waymo_motion.py:246-266`); drive it with :meth:`forward`.  `features` is the pre-processed scene merged
        with the encoder outputs (un-repeated: K futures share scene tensors, instance n uses scene n // K);
        `latent` / `goal` / `goal_valid` are per instance [N, ...] as in the reference."""
        if not deterministic_action:
            raise NotImplementedError("stochastic actions (training rollouts) are outside the built path")
        if gt_sdc is not None or require_vis_dict:
            raise NotImplementedError("what-if (gt_sdc) / visualisation dicts are outside the built path")
        if step_start != self.hparams["time_step_sim_start"]:
            raise NotImplementedError("step_start must equal time_step_sim_start")
        z = latent.sample(deterministic_latent, eps=latent_eps)
        b = features["agent_valid"].shape[0]
        mean_scene = latent.mean.reshape(b, k_futures, *latent.mean.shape[1:])[:, 0].contiguous()
        # traffic_rule_checker.enable_check_* (traffic_bots.yaml:240-244): the flag-gated checks do not feed back into the
        # simulation, they are evaluated on the recorded per-step states once the rollout is enqueued (tb_rule_checks)
        flags = self.hparams.get("traffic_rule_checker", {})
        want_checks = any(bool(v) for k, v in flags.items() if k.startswith("enable_check_"))
        out = self.engine.rollout(
            features, features, z, mean_scene, goal, goal_valid, k_futures, step_end,
            mask_teacher_forcing=mask_teacher_forcing.to(torch.uint8).contiguous(), tap_step=tap_step, stepwise=stepwise,
            record_check_states=want_checks,
        )
        self._step_t = step_start
        self._rollout_meta = (step_start, step_end, z)
        self._rule_ctx = (features, k_futures, flags) if want_checks else None
        if want_checks and not stepwise:
            out.update(self.engine.rule_checks(features, out["check_state"], out["check_valid"], k_futures, flags))
        return self._buffer_from(out)

    def _buffer_from(self, out: Dict[str, Tensor]) -> RolloutBuffer:
        step_start, step_end, z = self._rollout_meta
        buf = RolloutBuffer(step_start, step_end, self.hparams["time_step_current"])
        buf.valid = out["valid"].bool()
        buf.preds = out["preds"]
        buf.override_masks = out["override_masks"].bool()
        zeros = torch.zeros_like(buf.valid)
        buf.violations = {k: (out[k].bool() if k in out else zeros) for k in _VIOLATION_KEYS}
        n_step = buf.valid.shape[2]
        buf.latent_log_probs = out["latent_log_prob"].unsqueeze(-1).expand(-1, -1, n_step)
        buf.action_log_probs = out["action_log_probs"]
        buf.final = {k: out[k] for k in ("final_state", "final_valid", "final_hidden")}
        buf.taps = {k: out[k] for k in ("tap_policy_feature", "tap_agent_feature") if k in out}
        buf.latent_sample = z
        return buf

    def finish_rollout(self) -> RolloutBuffer:
        """`RolloutBuffer.finish()` for a stepwise rollout (`buffer.py:72-90`): the buffer over the steps taken so far
        (slots of steps not yet simulated are undefined)."""
        out = self.engine._step_out
        if getattr(self, "_rule_ctx", None) is not None:
            features, k_futures, flags = self._rule_ctx
            out.update(self.engine.rule_checks(features, out["check_state"], out["check_valid"], k_futures, flags))
        return self._buffer_from(out)

    def forward(self, *unused_feature_args, action_override=None, mask_action_override=None, state_override=None,
                mask_state_override=None, deterministic_action: bool = True, require_train_dict: bool = True,
                require_vis_dict: bool = False):
        """One simulation step, the reference's stateful `WaymoMotion.forward` (`waymo_motion.py:108-203`): advances the
        simulator opened by `rollout(..., stepwise=True)` and returns `(agent_state, agent_valid, train_dict, vis_dict)`.
        The scene / goal features and the teacher-forcing overrides were bound when the rollout was opened (the reference's
        `rollout()` passes exactly those per step, `waymo_motion.py:271-306`); custom per-call overrides are not built."""
        if action_override is not None or state_override is not None or mask_state_override is not None or require_vis_dict:
            raise NotImplementedError("per-call action/state overrides and vis dicts are outside the built path")
        if not deterministic_action:
            raise NotImplementedError("stochastic actions are outside the built path")
        eng = self.engine
        eng.rollout_step()
        st = eng.rollout_state()
        s_idx = self._step_t - self.hparams["time_step_sim_start"]
        self._step_t += 1
        o = eng._step_out
        train_dict = {}
        if require_train_dict:
            train_dict = {
                "latent_log_prob": o["latent_log_prob"], "action_log_prob": o["action_log_probs"][:, :, s_idx],
                "pred_valid": o["valid"][:, :, s_idx].bool(), "pred_state": o["preds"][:, :, s_idx],
            }
        self.model.hidden = st["hidden"].flatten(1, 2)
        return st["agent_state"], st["agent_valid"].bool(), train_dict, {}

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
    ) -> Tuple[RolloutBuffer, Tensor, Tensor]:
        """`WaymoMotion.joint_future_pred` (`waymo_motion.py:478-572`): K futures per scene, sample 0
        deterministic.  `batch` is the pre-processed scene.  The reference materialises every tensor K times
        with repeat_interleave; here only the per-instance tensors are."""
        k = self.hparams["n_joint_future"]
        b, _, a = batch["agent_valid"].shape
        deterministic = torch.zeros(b * k, a, dtype=torch.bool, device=self.device)
        deterministic[::k] = True
        latent.repeat_interleave_(k, 0)
        goal.repeat_interleave_(k, 0)
        if goal_sample is None:
            goal_sample = goal.sample(deterministic, generator=generator)
        else:
            goal_sample = goal_sample.to(self.device).reshape(b * k, a)
        goal_log_probs = goal.log_prob(goal_sample)
        goal_valid = goal_valid.repeat_interleave(k, 0)
        features = dict(batch)
        features["map_feature"] = input_feature_dict["map_feature"]
        features["map_feature_valid"] = input_feature_dict["map_feature_valid"].to(torch.uint8).contiguous()
        features["tl_feature"] = input_feature_dict["tl_feature"]
        tf = self.hparams["teacher_forcing_joint_future_pred"]
        mask_tf = teacher_forcing_mask(batch["agent_valid"].bool(), tf.get("step_spawn_agent", 10), tf.get("step_warm_start", 10))
        buf = self.rollout(
            features, latent=latent, goal=goal_sample, goal_valid=goal_valid, mask_teacher_forcing=mask_tf,
            deterministic_latent=deterministic, deterministic_action=True,
            step_start=self.hparams["time_step_sim_start"], step_end=self.hparams["time_step_end"],
            k_futures=k, latent_eps=latent_eps, tap_step=tap_step,
        )
        buf.flatten_repeat(k)
        goal_log_probs = goal_log_probs.view(b, k, a).transpose(1, 2)
        goal_sample = goal_sample.view(b, k, a).transpose(1, 2)
        return buf, goal_sample, goal_log_probs

    def test_step(self, batch: Dict[str, Tensor], batch_idx: int = 0, latent_eps=None, goal_sample=None, generator=None,
                  tap_step: int = -1) -> Dict[str, Tensor]:
        """`WaymoMotion.test_step` (`waymo_motion.py:902-940`) up to and including `waymo_post_processing`; the submission writer
        (`:942-949`) is out of scope.  Returns the buffer, the intermediate products and the post-processed `pred_dict`."""
        scene = self.pre_processing(batch)
        input_feature_dict = self.model.encode_input_features(scene)
        goal_valid = scene["agent_valid"].bool().any(1)
        goal_pred = self.model.goal_manager.pred_goal()
        latent_prior = self.model.latent_encoder()
        latent_mean, latent_valid = latent_prior.mean, latent_prior.valid  # (repeat_interleave_ below rebinds them)
        buf, gs, glp = self.joint_future_pred(
            scene, input_feature_dict, latent_prior, goal_pred, goal_valid, latent_eps=latent_eps, goal_sample=goal_sample,
            generator=generator, tap_step=tap_step,
        )
        scores = torch.exp(buf.latent_log_probs[..., 0] + glp)  # waymo_motion.py:936
        pred_dict = None
        if buf.preds.shape[3] > buf.step_future_start:  # (a rollout that stops at the current step has no future to post-process)
            pred_dict = self.waymo_post_processing(
                valid=buf.valid[:, :, 0].any(-1), scores=scores, trajs=buf.preds[:, :, :, buf.step_future_start:],
                agent_type=scene["agent_type"],
            )
        return {
            "rollout_buffer": buf, "goal_sample": gs, "goal_log_probs": glp, "input_feature_dict": input_feature_dict,
            "latent_mean": latent_mean, "latent_valid": latent_valid, "dest_logits": self.model._enc["dest_logits"],
            "scores": scores, "pred_dict": pred_dict,
        }
