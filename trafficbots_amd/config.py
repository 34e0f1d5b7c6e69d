"""Config surface of the rollout hot path.

The reference composes its model config with Hydra/OmegaConf
(`configs/model/traffic_bots.yaml:1-245`, instantiated at `src/run.py:34-36`).
Neither package is needed here: this module holds the default values of every key
the hot path reads (same key names, same nesting) as a plain dict, a loader for
yaml files of that shape (PyYAML) and a resolver for OmegaConf's relative
interpolations (`${..hidden_dim}`, `${.map}`), so an existing reference config
file can be passed unchanged.

Branches of the reference that are not built (ablations listed in
`docs/ablation_models.md`) are rejected loudly by :func:`check_supported`.
"""
from __future__ import annotations

import copy
import re
from typing import Any, Dict

# Defaults for the keys of configs/model/traffic_bots.yaml that the path reads.
# Training-only groups (optimizer, lr_scheduler, sub_womd_*) are accepted by the loader but ignored;
# training_metrics and differentiable_reward feed the forward losses of validation_step.
_MLP_CFG = {"use_layernorm": False, "activation": "relu", "dropout_p": 0.1}

DEFAULT_MODEL_CONFIG: Dict[str, Any] = {
    "time_step_current": 10,
    "time_step_gt": 90,
    "time_step_end": 90,
    "time_step_sim_start": 1,
    "hidden_dim": 128,
    "n_joint_future": 6,
    "detach_state_policy": True,
    "pre_processing": {
        "input": {
            "dropout_p_history": -1,
            "pe_dim": 96,
            "pose_pe": {"map": "pe_xy_yaw", "tl": "pe_xy_yaw", "agent": "pe_xy_yaw"},
        },
        # SceneCentricLatent (sc_latent.py): only the posterior-side history dropout is read (train mode, explicit draws);
        # (perturb_input_to_latent: built since round 5 for training_step, WaymoMotion._perturb_latent_inputs)
        "latent": {"perturb_input_to_latent": False, "dropout_p_history": -1, "max_meter": 50.0, "max_rad": 3.14},
    },
    "p_drop_hidden": -1.0,
    "model": {
        "hidden_dim": 128,
        "add_goal_latent_first": False,
        "resample_latent": False,
        "n_layer_tf_as2pl": 3,
        "n_layer_tf_as2tl": 3,
        "tf_cfg": {
            "d_model": 128,
            "n_head": 4,
            "dropout_p": 0.1,
            "norm_first": True,
            "bias": True,
            "activation": "relu",
            "d_feedforward": 128,
            "out_layernorm": False,
        },
        "input_pe_encoder": {
            "pe_mode": "cat",
            "n_layer": 2,
            "mlp_dropout_p": 0.1,
            "mlp_use_layernorm": False,
        },
        "map_encoder": {
            "pool_mode": "max",
            "densetnt_vectornet": True,
            "n_layer": 3,
            "mlp_dropout_p": 0.1,
            "mlp_use_layernorm": False,
        },
        "goal_manager": {
            "disable_if_reached": True,
            "goal_predictor": {
                "mode": "mlp",
                "n_layer_gru": 3,
                "use_layernorm": True,
                "res_add_gru": True,
                "detach_features": True,
            },
            "goal_attr_mode": "dest",
            "goal_in_local": True,
            "dest_detach_map_feature": False,
        },
        "latent_encoder": {
            "latent_dim": 16,
            "temporal_down_sample_rate": 5,
            "shared_post_prior_net": False,
            "shared_transformer_as": True,
            "latent_prior": {"dist_type": "diag_gaus", "n_cat": 8, "log_std": -1, "use_layernorm": False},
            "latent_post": {"dist_type": "diag_gaus", "n_cat": 8, "log_std": -1, "use_layernorm": False},
        },
        "temporal_aggregate": {"mode": "max_valid"},
        "agent_temporal": {
            "_target_": "models.modules.agent_temporal.MultiAgentGRULoop",
            "num_layers": 3,
            "dropout": 0.1,
        },
        "agent_interaction": {
            "n_layer": 3,
            "mask_self_agent": True,
            "detach_tgt": False,
            "attn_to_map_aware_feature": True,
        },
        "add_latent": {
            "mode": "cat",
            "res_cat": False,
            "res_add": True,
            "n_layer_mlp_in": 2,
            "n_layer_mlp_out": 2,
            "mlp_in_cfg": dict(_MLP_CFG),
            "mlp_out_cfg": dict(_MLP_CFG),
        },
        "add_goal": {
            "mode": "cat",
            "res_cat": False,
            "res_add": True,
            "n_layer_mlp_in": 3,
            "n_layer_mlp_out": 2,
            "mlp_in_cfg": {"use_layernorm": True, "activation": "relu", "dropout_p": 0.1},
            "mlp_out_cfg": dict(_MLP_CFG),
        },
        "interaction_first": True,
        "n_layer_final_mlp": -1,
    },
    "action_head": {"log_std": -2, "branch_type": True, "use_layernorm": False},
    "dynamics": {
        "use_veh_dynamics_for_all": False,
        "veh": {"_target_": "utils.dynamics.MultiPathPP", "max_acc": 5, "max_yaw_rate": 1.5, "disable_neg_spd": False},
        "cyc": {"_target_": "utils.dynamics.MultiPathPP", "max_acc": 6, "max_yaw_rate": 3, "disable_neg_spd": False},
        "ped": {"_target_": "utils.dynamics.MultiPathPP", "max_acc": 7, "max_yaw_rate": 7},
    },
    "teacher_forcing_training": {  # traffic_bots.yaml:127-133
        "step_spawn_agent": 10, "step_warm_start": 10, "step_horizon": 0, "step_horizon_decrease_per_epoch": 0,
        "prob_forcing_agent": 0, "prob_forcing_agent_decrease_per_epoch": 0,
    },
    "p_training_rollout_prior": 0.1,  # traffic_bots.yaml
    "training_deterministic_action": True,
    "teacher_forcing_joint_future_pred": {"step_spawn_agent": 10, "step_warm_start": 10},
    "teacher_forcing_reactive_replay": {"step_spawn_agent": 90, "step_warm_start": 10},
    "waymo_post_processing": {  # traffic_bots.yaml:179-186
        "k_pred": 6, "use_ade": True, "score_temperature": 1e2, "mpa_nms_thresh": [], "mtr_nms_thresh": [], "aggr_thresh": [],
        "n_iter_em": 3,
    },
    "differentiable_reward": {  # traffic_bots.yaml:157-171
        "w_collision": 0, "reduce_collsion_with_max": True, "use_il_loss": True,
        "l_pos": {"weight": 1e-1, "criterion": "SmoothL1Loss"},
        "l_rot": {"weight": 1e1, "criterion": "SmoothL1Loss", "angular_type": "cosine"},
        "l_spd": {"weight": 1e-1, "criterion": "SmoothL1Loss"},
    },
    "training_metrics": {  # traffic_bots.yaml:208-219
        "w_vae_kl": 1e-1, "kl_balance_scale": -1, "kl_free_nats": 1e-2, "kl_for_unseen_agent": True, "w_diffbar_reward": 1.0,
        "w_goal": 1.0, "w_relevant_agent": 0, "p_loss_for_irrelevant": -1.0, "loss_for_teacher_forcing": True,
        "step_training_start": 10,
    },
    "traffic_rule_checker": {
        "enable_check_collided": False,
        "enable_check_run_road_edge": False,
        "enable_check_run_red_light": False,
        "enable_check_passive": False,
    },
}

_INTERP = re.compile(r"^\$\{(\.+)([A-Za-z0-9_.]+)\}$")


def _lookup(root: Dict[str, Any], path: list) -> Any:
    node = root
    for p in path:
        node = node[p]
    return node


def _has_interp(node: Any) -> bool:
    if isinstance(node, str):
        return _INTERP.match(node.strip()) is not None
    if isinstance(node, dict):
        return any(_has_interp(v) for v in node.values())
    if isinstance(node, list):
        return any(_has_interp(v) for v in node)
    return False


def resolve_interpolations(cfg: Dict[str, Any]) -> Dict[str, Any]:
    """Resolve OmegaConf relative interpolations in a nested dict.

    `${.x}` names a sibling of the key that holds the string and every extra dot
    climbs one level (the form used by `configs/model/traffic_bots.yaml:24,28,38`).
    A string is replaced only once its target subtree is itself free of
    interpolations, so copied sub-dicts never carry relative references to a new place;
    the tree is swept until nothing changes.
    """
    cfg = copy.deepcopy(cfg)

    def sweep(node: Dict[str, Any], parents: list) -> int:
        n_done = 0
        for k in list(node.keys()):
            v = node[k]
            if isinstance(v, dict):
                n_done += sweep(v, parents + [k])
            elif isinstance(v, str):
                m = _INTERP.match(v.strip())
                if m is None:
                    continue
                ups = len(m.group(1)) - 1
                base = parents[: len(parents) - ups] if ups else list(parents)
                target = _lookup(cfg, base + m.group(2).split("."))
                if not _has_interp(target):
                    node[k] = copy.deepcopy(target)
                    n_done += 1
        return n_done

    for _ in range(32):
        if sweep(cfg, []) == 0:
            break
    if _has_interp(cfg):
        raise ValueError("unresolvable (cyclic or absolute) interpolation in config")
    return cfg


def load_model_config(path: str | None = None, overrides: Dict[str, Any] | None = None) -> Dict[str, Any]:
    """Default config, optionally replaced by a yaml file of the reference's shape
    and patched with dotted-key overrides (`{"model.n_layer_tf_as2pl": 3}`)."""
    if path is None:
        cfg = copy.deepcopy(DEFAULT_MODEL_CONFIG)
    else:
        import yaml

        with open(path, "r") as f:
            cfg = yaml.safe_load(f)
        cfg = resolve_interpolations(cfg)
    for key, val in (overrides or {}).items():
        node = cfg
        parts = key.split(".")
        for p in parts[:-1]:
            node = node.setdefault(p, {})
        node[parts[-1]] = val
    for group in ("differentiable_reward", "training_metrics", "teacher_forcing_training", "p_training_rollout_prior",
                  "training_deterministic_action"):
        cfg.setdefault(group, copy.deepcopy(DEFAULT_MODEL_CONFIG[group]))
    check_supported(cfg)
    return cfg


def _plain(node: Any) -> Any:
    """dict-likes (OmegaConf DictConfig, attribute dicts) and list-likes -> plain dict / list, recursively"""
    if hasattr(node, "items"):
        return {str(k): _plain(v) for k, v in node.items()}
    if isinstance(node, (list, tuple)) or (hasattr(node, "__iter__") and not isinstance(node, (str, bytes)) and hasattr(node, "__len__")
                                              and not hasattr(node, "shape")):
        return [_plain(v) for v in node]
    return node


def config_from_hydra_kwargs(kwargs: Dict[str, Any]) -> Dict[str, Any]:
    """The keyword arguments `hydra.utils.instantiate` hands `WaymoMotion.__init__` for `configs/model/traffic_bots.yaml`
    (`src/pl_modules/waymo_motion.py:28-62`, `src/run.py:34-36`) -> this package's config dict: same keys, `_target_` entries
    kept where the supported-branch check reads them, interpolations resolved if the caller did not, groups the hot path does not
    read (data_size, optimizer, lr_scheduler, sub_womd_*, n_video_batch, wb_artifact, step_detach_hidden, p_drop_hidden, lr_goal,
    interactive_challenge) accepted and carried along untouched.  Missing OPTIONAL groups get this package's defaults; the
    operand_precision extension key is honoured."""
    cfg = _plain(kwargs)
    cfg.pop("_target_", None)
    cfg = resolve_interpolations(cfg)
    for key, val in DEFAULT_MODEL_CONFIG.items():
        if key not in cfg:
            if key in ("model", "dynamics", "action_head", "pre_processing"):
                raise KeyError(f"WaymoMotion(**kwargs): the reference's constructor argument '{key}' is missing")
            cfg[key] = copy.deepcopy(val)
    # (the reference's pre_processing group also lists scene_centric / latent stages, `traffic_bots.yaml:13-32`; only `input` is read)
    cfg["pre_processing"].setdefault("input", copy.deepcopy(DEFAULT_MODEL_CONFIG["pre_processing"]["input"]))
    check_supported(cfg)
    return cfg


REFERENCE_TARGETS = ("pl_modules.waymo_motion.WaymoMotion", "trafficbots_amd.waymo_motion.WaymoMotion", "trafficbots_amd.WaymoMotion")


def instantiate(cfg: Dict[str, Any], **kwargs):
    """`hydra.utils.instantiate(cfg.model, ...)` for this package (`src/run.py:34-36`): `cfg` is a dict shaped like
    `configs/model/traffic_bots.yaml` -- `_target_: pl_modules.waymo_motion.WaymoMotion` (or this package's class path) plus the
    constructor arguments; `kwargs` are added on top (e.g. data_size=..., device="cuda:1").  Nested `_target_`s (the model, the
    pre-processing stages, the dynamics) are not instantiated one by one: `WaymoMotion` takes the groups as configuration."""
    from .waymo_motion import WaymoMotion

    cfg = _plain(cfg)
    target = cfg.pop("_target_", REFERENCE_TARGETS[0])
    if target not in REFERENCE_TARGETS:
        raise NotImplementedError(f"trafficbots_amd.instantiate: _target_ '{target}' is not the hot path's task module")
    cfg.update(kwargs)
    return WaymoMotion(**cfg)


def check_supported(cfg: Dict[str, Any]) -> None:
    """Raise NotImplementedError for config branches outside the built path."""
    m = cfg["model"]

    def need(cond: bool, what: str) -> None:
        if not cond:
            raise NotImplementedError(f"trafficbots_amd: unsupported config branch: {what}")

    need(cfg["hidden_dim"] == 128 and m["tf_cfg"]["d_model"] == 128, "hidden_dim != 128")
    need(m["tf_cfg"]["n_head"] == 4, "tf_cfg.n_head != 4")
    need(m["tf_cfg"]["norm_first"] is True, "tf_cfg.norm_first=False")
    need(m["tf_cfg"]["d_feedforward"] == 128, "tf_cfg.d_feedforward != 128")
    need(m["tf_cfg"]["activation"] == "relu", "tf_cfg.activation != relu")
    need(not m["tf_cfg"].get("out_layernorm", False), "tf_cfg.out_layernorm")
    need(m["tf_cfg"].get("bias", True), "tf_cfg.bias=False")
    need(m["input_pe_encoder"]["pe_mode"] == "cat", "input_pe_encoder.pe_mode != cat")
    need(m["input_pe_encoder"]["n_layer"] == 2, "input_pe_encoder.n_layer != 2")
    need(not m["input_pe_encoder"]["mlp_use_layernorm"], "input_pe_encoder.mlp_use_layernorm")
    need(cfg["pre_processing"]["input"]["pe_dim"] == 96, "pre_processing.input.pe_dim != 96")
    for k in ("map", "tl", "agent"):
        need(cfg["pre_processing"]["input"]["pose_pe"][k] == "pe_xy_yaw", f"pose_pe.{k} != pe_xy_yaw")
    need(m["map_encoder"]["densetnt_vectornet"] and m["map_encoder"]["pool_mode"] == "max", "map_encoder mode")
    need(m["map_encoder"]["n_layer"] == 3, "map_encoder.n_layer != 3")
    need(m["n_layer_tf_as2pl"] == 3 and m["n_layer_tf_as2tl"] == 3, "n_layer_tf_as2pl/as2tl != 3")
    gm = m["goal_manager"]
    need(gm["goal_attr_mode"] == "dest", "goal_manager.goal_attr_mode != dest")
    need(gm["goal_predictor"]["mode"] == "mlp", "goal_predictor.mode != mlp")
    need(gm["goal_predictor"]["n_layer_gru"] == 3, "goal_predictor.n_layer_gru != 3")
    need(gm["goal_predictor"]["use_layernorm"] and gm["goal_predictor"]["res_add_gru"], "goal_predictor flags")
    need(gm["disable_if_reached"], "goal_manager.disable_if_reached=False")
    le = m["latent_encoder"]
    need(le["latent_dim"] == 16, "latent_encoder.latent_dim != 16")
    need(le["latent_prior"]["dist_type"] == "diag_gaus", "latent_prior.dist_type != diag_gaus")
    need(le["latent_prior"]["log_std"] is not None, "latent_prior.log_std=None")
    need(not le["latent_prior"]["use_layernorm"], "latent_prior.use_layernorm")
    need(le["shared_transformer_as"] and not le["shared_post_prior_net"], "latent_encoder sharing flags")
    need(le["temporal_down_sample_rate"] == 5, "temporal_down_sample_rate != 5")
    need(m["temporal_aggregate"]["mode"] == "max_valid", "temporal_aggregate.mode != max_valid")
    need(m["agent_temporal"]["_target_"].endswith("MultiAgentGRULoop"), "agent_temporal._target_")
    need(m["agent_temporal"]["num_layers"] == 3, "agent_temporal.num_layers != 3")
    ai = m["agent_interaction"]
    need(ai["n_layer"] == 3 and ai["mask_self_agent"] and ai["attn_to_map_aware_feature"], "agent_interaction flags")
    for name, n_in in (("add_latent", 2), ("add_goal", 3)):
        a = m[name]
        need(a["mode"] == "cat" and a["res_add"] and not a["res_cat"], f"{name}.mode/res flags")
        need(a["n_layer_mlp_in"] == n_in and a["n_layer_mlp_out"] == 2, f"{name}.n_layer_mlp_*")
        need(a["mlp_in_cfg"]["activation"] == "relu" and a["mlp_out_cfg"]["activation"] == "relu", f"{name} act")
        need(not a["mlp_out_cfg"]["use_layernorm"], f"{name}.mlp_out_cfg.use_layernorm")
    need(m["add_goal"]["mlp_in_cfg"]["use_layernorm"], "add_goal.mlp_in_cfg.use_layernorm=False")
    need(not m["add_latent"]["mlp_in_cfg"]["use_layernorm"], "add_latent.mlp_in_cfg.use_layernorm=True")
    need(m["interaction_first"] and not m["add_goal_latent_first"], "interaction_first/add_goal_latent_first")
    need(not m["resample_latent"], "resample_latent=True")
    need(m["n_layer_final_mlp"] <= 0, "n_layer_final_mlp > 0")
    ah = cfg["action_head"]
    need(ah["branch_type"] and ah["log_std"] is not None and not ah["use_layernorm"], "action_head flags")
    dy = cfg["dynamics"]
    need(not dy["use_veh_dynamics_for_all"], "dynamics.use_veh_dynamics_for_all")
    for k in ("veh", "cyc", "ped"):
        need(dy[k]["_target_"].endswith("MultiPathPP"), f"dynamics.{k}._target_ != MultiPathPP")
        need(not dy[k].get("disable_neg_spd", False), f"dynamics.{k}.disable_neg_spd")
    # traffic_rule_checker.enable_check_*: all four flag-gated checks are built (tb_rule_checks)
