"""CPU restatement of the reference's forward training losses -- TEST INFRASTRUCTURE ONLY.

Follows `src/utils/rewards.py:33-131` (`DifferentiableReward.get`), `src/models/metrics/loss.py:9-32,35-74` (`AngularError`,
`BalancedKL`) and `src/models/metrics/training.py:62-139` (`TrainingMetrics.update`) of zhejz/TrafficBots, evaluated over a
recorded rollout (the reward is a pure function of each step's pre-override prediction and the ground truth of that step).
Pinned by tests/golden/val_*.npz (tools/gen_golden_val.py), which hold what the imported reference computed.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg may import this module.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
from torch import Tensor


def _smooth_l1(a: Tensor, b: Tensor) -> Tensor:
    d = (a - b).abs()
    return torch.where(d < 1.0, 0.5 * d * d, d - 0.5)


def _mse(a: Tensor, b: Tensor) -> Tensor:
    return (a - b) ** 2


def _l1(a: Tensor, b: Tensor) -> Tensor:
    return (a - b).abs()


CRITERIA = {"SmoothL1Loss": _smooth_l1, "MSELoss": _mse, "L1Loss": _l1}


def _cast_rad(x: Tensor) -> Tensor:
    return (x + math.pi) % (2 * math.pi) - math.pi


def angular_error(criterion: str, angular_type: Optional[str], preds: Tensor, target: Tensor) -> Tensor:
    """`AngularError.compute` (`loss.py:17-32`)."""
    if angular_type is None:
        return CRITERIA[criterion](preds, target)
    if angular_type == "cast":
        d = _cast_rad(preds - target)
        return CRITERIA[criterion](d, torch.zeros_like(d))
    if angular_type == "cosine":
        return 0.5 * (1 - torch.cos(preds - target))
    if angular_type == "vector":
        return CRITERIA[criterion](torch.cos(preds), torch.cos(target)) + CRITERIA[criterion](torch.sin(preds), torch.sin(target))
    raise ValueError(angular_type)


def differentiable_reward(pred_valid: Tensor, pred_state: Tensor, gt_valid: Optional[Tensor], gt_state: Optional[Tensor],
                          agent_size: Tensor, cfg: dict):
    """pred_valid [N,A,S], pred_state [N,A,S,4], gt_valid [N,A,S] / gt_state [N,A,S,4] (or None past the ground truth),
    agent_size [N,A,3]; cfg = the `differentiable_reward` config group.  Returns (reward [N,A,S], reward_valid [N,A,S])."""
    reward = torch.zeros_like(pred_state[..., 0])
    reward_valid = pred_valid
    w_col = float(cfg["w_collision"])
    if w_col > 0:  # rewards.py:50-113, five circles per box
        n, a, s_len = pred_valid.shape
        eps = torch.finfo(pred_state.dtype).eps
        xy = pred_state[..., :2]
        head = torch.stack([torch.cos(pred_state[..., 2]), torch.sin(pred_state[..., 2])], -1)
        w = agent_size[:, :, :2].amin(-1)
        l = agent_size[:, :, :2].amax(-1)
        d = ((l - w) / 4.0)[:, :, None, None]
        cen = torch.stack([xy + k * head * d for k in (-2, -1, 0, 1, 2)], -2)  # [N,A,S,5,2]
        c0 = cen[:, :, None, :, :, None, :]  # agent i, circle p
        c1 = cen[:, None, :, :, None, :, :]  # agent j, circle q
        dist = torch.norm(c0 - c1, dim=-1) + eps  # [N,Ai,Aj,S,5,5]
        dist = dist.flatten(-2).amin(-1)
        r = w / 2.0 + eps
        r_sum = (r[:, :, None] + r[:, None, :])[..., None]
        col = torch.clamp(1 - dist / r_sum, min=0)
        ego = torch.eye(a, dtype=torch.bool)[None, :, :, None] | ~pred_valid[:, :, None, :] | ~pred_valid[:, None, :, :]
        col = col.masked_fill(ego, 0.0)
        if cfg["reduce_collsion_with_max"]:
            col = col.amax(2)
        else:
            col = torch.clamp(col, max=1).sum(2) / pred_valid.sum(1, keepdim=True)
        reward = reward - w_col * col.masked_fill(~pred_valid, 0.0)
    if cfg["use_il_loss"] and gt_valid is not None:  # rewards.py:115-129
        inv = ~(pred_valid & gt_valid).unsqueeze(-1)
        g = gt_state.masked_fill(inv, 0)
        p = pred_state.masked_fill(inv, 0)
        e_pos = CRITERIA[cfg["l_pos"]["criterion"]](g[..., :2], p[..., :2]).sum(-1)
        e_rot = angular_error(cfg["l_rot"]["criterion"], cfg["l_rot"]["angular_type"], g[..., 2], p[..., 2])
        e_spd = CRITERIA[cfg["l_spd"]["criterion"]](g[..., 3], p[..., 3])
        reward = reward - (cfg["l_pos"]["weight"] * e_pos + cfg["l_rot"]["weight"] * e_rot + cfg["l_spd"]["weight"] * e_spd)
        reward_valid = pred_valid & gt_valid
    return reward.masked_fill(~reward_valid, 0.0), reward_valid


def diag_gaussian_kl(q_mean: Tensor, q_log_std: Tensor, p_mean: Tensor, p_log_std: Tensor) -> Tensor:
    """kl_divergence(Independent(Normal q), Independent(Normal p)) summed over the event dim (torch/distributions/kl.py
    `_kl_normal_normal`)."""
    var_ratio = (q_log_std.exp() / p_log_std.exp()) ** 2
    t1 = ((q_mean - p_mean) / p_log_std.exp()) ** 2
    return (0.5 * (var_ratio + t1 - 1 - var_ratio.log())).sum(-1)


def training_metric_states(pred_valid: Tensor, reward_valid: Tensor, reward: Tensor, override_masks: Tensor, agent_role: Tensor,
                           dest_logits: Tensor, goal_dist_valid: Tensor, gt_dest: Tensor, post_mean: Tensor,
                           post_log_std: Tensor, post_valid: Tensor, prior_mean: Tensor, prior_log_std: Tensor,
                           prior_valid: Tensor, cfg: dict, irrelevant_draw: Optional[Tensor] = None) -> Dict[str, float]:
    """`TrainingMetrics.update` (`training.py:62-139`) for one batch, starting from zeroed states.
    pred_valid / reward_valid / reward / override_masks: [B,A,S]; dest_logits [B,A,P] masked, un-normalised; goal_dist_valid =
    `goal_pred.valid`; cfg = the `training_metrics` group.  `irrelevant_draw` [B,A] = the Bernoulli(p_loss_for_irrelevant) draw the
    reference takes from torch's stream (`:88`), passed explicitly."""
    pv = pred_valid.clone()
    if cfg["p_loss_for_irrelevant"] > 0:
        assert irrelevant_draw is not None
        pv = (pv & agent_role.any(-1).unsqueeze(-1)) | irrelevant_draw.bool().reshape(*pv.shape[:2], 1)
    if not cfg["loss_for_teacher_forcing"]:
        pv = pv & ~override_masks
    if cfg["step_training_start"] > 0:
        pv[:, :, : cfg["step_training_start"]] = False
    w_rel = None
    if cfg["w_relevant_agent"] > 0:
        w_rel = pv.any(-1).to(reward.dtype) + agent_role.any(-1).to(reward.dtype) * cfg["w_relevant_agent"]
    out: Dict[str, float] = {}
    if cfg["w_vae_kl"] > 0:
        kv = (post_valid if cfg["kl_for_unseen_agent"] else prior_valid) & pv.any(-1)
        free = cfg["kl_free_nats"]
        alpha = cfg["kl_balance_scale"]
        kl = diag_gaussian_kl(post_mean, post_log_std.expand_as(post_mean), prior_mean, prior_log_std.expand_as(prior_mean))
        if alpha > 0:  # forward value of KL balancing: both terms equal the plain KL (detach only changes gradients)
            k0 = torch.clamp(kl, min=free) if free > 0 else kl
            err = alpha * k0 + (1 - alpha) * k0
        else:
            err = torch.clamp(kl, min=free) if free > 0 else kl
        if w_rel is not None:
            err = err * w_rel
        out["vae_kl_counter"] = float(kv.sum())
        out["vae_kl"] = float(err.masked_fill(~kv, 0.0).sum())
    if cfg["w_diffbar_reward"] > 0:
        rv = pv & reward_valid
        r = reward.masked_fill(~rv, 0.0)
        if w_rel is not None:  # training.py:124 multiplies [B,A,S] by w_mask_rel.unsqueeze(1): only well-formed for A == S
            r = r * w_rel.unsqueeze(1)
        out["diffbar_reward"] = float(-r.sum())
        out["diffbar_reward_counter"] = float(rv.sum())
    if cfg["w_goal"] > 0:
        gv = goal_dist_valid & pv.any(-1)
        logp = torch.log_softmax(dest_logits, -1).gather(-1, gt_dest.unsqueeze(-1)).squeeze(-1)
        nll = (-logp).masked_fill(~gv, 0)
        if w_rel is not None:
            nll = nll * w_rel
        out["goal_loss"] = float(nll.sum())
        out["goal_counter"] = float(gv.sum())
    return out


def training_metric_compute(st: Dict[str, float], cfg: dict, prefix: str) -> Dict[str, float]:
    """`TrainingMetrics.compute` (`training.py:141-158`)."""
    out = {f"{prefix}/loss": 0.0}
    if cfg["w_vae_kl"] > 0:
        out[f"{prefix}/vae_kl"] = cfg["w_vae_kl"] * st["vae_kl"] / st["vae_kl_counter"]
        out[f"{prefix}/loss"] += out[f"{prefix}/vae_kl"]
    if cfg["w_diffbar_reward"] > 0:
        out[f"{prefix}/diffbar_reward"] = cfg["w_diffbar_reward"] * st["diffbar_reward"] / st["diffbar_reward_counter"]
        out[f"{prefix}/loss"] += out[f"{prefix}/diffbar_reward"]
    if cfg["w_goal"] > 0:
        out[f"{prefix}/goal_loss"] = cfg["w_goal"] * st["goal_loss"] / st["goal_counter"]
        out[f"{prefix}/loss"] += out[f"{prefix}/goal_loss"]
    return out
