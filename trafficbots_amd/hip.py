"""ctypes binding of `libtrafficbots_hip.so` (C ABI: include/trafficbots_hip.h).

There is NO CPU fallback: importing this module without the built library, or creating a
context without a visible HIP device, raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional

# TB_HIP_LIB selects an alternative build of the same library (e.g. the -DTB_PROFILE one used by tools/)
_LIB_PATH = os.environ.get("TB_HIP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libtrafficbots_hip.so")

c_f32p = C.POINTER(C.c_float)
c_u8p = C.POINTER(C.c_uint8)
c_i32p = C.POINTER(C.c_int32)


SWITCH_FIELDS = ("step_helpers", "step_l2_warmers", "step_pre_inter", "step_w3", "step_aw", "step_lean", "rollout_graph", "encode_pack",
                 "encode_side", "encode_dest_side", "dest_lds_pad")


class TbSwitches(C.Structure):
    """`tb_switches` (include/trafficbots_hip.h): launch-shaping switches, every field 0 = automatic."""
    _fields_ = [(n, C.c_int32) for n in SWITCH_FIELDS] + [("reserved", C.c_int32 * 5)]


class TbConfig(C.Structure):
    _fields_ = [
        ("time_step_current", C.c_int32),
        ("time_step_sim_start", C.c_int32),
        ("dt", C.c_float),
        ("max_acc", C.c_float * 3),
        ("max_yaw_rate", C.c_float * 3),
        ("action_log_std", C.c_float),
        ("latent_log_std", C.c_float),
        ("operand_precision", C.c_int32),
        ("sw", TbSwitches),
    ]


class TbRolloutIO(C.Structure):
    _fields_ = [
        ("n_scene", C.c_int32), ("k_futures", C.c_int32), ("n_agent", C.c_int32), ("n_pl", C.c_int32),
        ("n_tl", C.c_int32), ("n_hist", C.c_int32), ("step_end", C.c_int32),
        ("map_feature", c_f32p), ("map_feature_valid", c_u8p), ("tl_feature", c_f32p), ("tl_feature_valid", c_u8p),
        ("agent_valid", c_u8p), ("agent_state", c_f32p), ("agent_vel", c_f32p), ("agent_acc", c_f32p),
        ("agent_yaw_rate", c_f32p), ("mask_teacher_forcing", c_u8p), ("agent_type", c_i32p), ("agent_size", c_f32p),
        ("map_boundary", c_f32p), ("map_valid", c_u8p), ("map_type", c_i32p), ("map_pos", c_f32p), ("map_dir", c_f32p),
        ("latent_sample", c_f32p), ("latent_mean", c_f32p), ("dest", c_i32p), ("goal_valid", c_u8p),
        ("preds", c_f32p), ("valid", c_u8p), ("override_masks", c_u8p), ("outside_map", c_u8p),
        ("outside_map_this_step", c_u8p), ("dest_reached", c_u8p), ("dest_reached_this_step", c_u8p),
        ("action_log_probs", c_f32p), ("latent_log_prob", c_f32p),
        ("final_state", c_f32p), ("final_valid", c_u8p), ("final_hidden", c_f32p),
        ("tap_step", C.c_int32), ("tap_policy_feature", c_f32p), ("tap_agent_feature", c_f32p),
        ("check_state", c_f32p), ("check_valid", c_u8p),
        ("n_tl_step", C.c_int32), ("latent_posterior", C.c_int32), ("warm_start_steps", C.c_int32),
        ("action_eps", c_f32p), ("hidden_drop", C.c_void_p),
        ("latent_eps", c_f32p), ("latent_deterministic", c_u8p), ("latent_sample_out", c_f32p), ("actions", c_f32p),
    ]


class TbLatentSampleIO(C.Structure):
    _fields_ = [("n_scene", C.c_int32), ("k_futures", C.c_int32), ("n_agent", C.c_int32), ("posterior", C.c_int32),
                ("log_std", c_f32p), ("mean", c_f32p), ("eps", c_f32p), ("deterministic", c_u8p), ("forced", c_f32p), ("sample", c_f32p), ("log_prob", c_f32p)]


class TbDestSampleIO(C.Structure):
    _fields_ = [("n_scene", C.c_int32), ("k_futures", C.c_int32), ("n_agent", C.c_int32), ("n_pl", C.c_int32), ("from_probs", C.c_int32),
                ("dest_logits", c_f32p), ("uniform", c_f32p), ("deterministic", c_u8p), ("forced", c_i32p), ("sample", c_i32p),
                ("log_prob", c_f32p), ("probs", c_f32p)]


class TbStepOverride(C.Structure):
    _fields_ = [("mask", c_u8p), ("agent_state", c_f32p), ("vel", c_f32p), ("acc", c_f32p), ("yaw_rate", c_f32p), ("gt_valid", c_u8p),
                ("action", c_f32p), ("action_mask", c_u8p)]


class TbForwardIO(C.Structure):
    _fields_ = [
        ("n_inst", C.c_int32), ("n_agent", C.c_int32), ("n_pl", C.c_int32), ("n_tl", C.c_int32),
        ("agent_valid", c_u8p), ("agent_feature", c_f32p), ("map_valid", c_u8p), ("map_feature", c_f32p),
        ("tl_valid", c_u8p), ("tl_feature", c_f32p), ("goal_valid", c_u8p), ("goal_feature", c_f32p),
        ("latent_sample", c_f32p), ("hidden", c_f32p), ("policy_feature", c_f32p),
        ("attn_pl", c_f32p), ("attn_tl", c_f32p), ("attn_agent", c_f32p),
    ]


class TbRuleIO(C.Structure):
    _fields_ = [
        ("n_scene", C.c_int32), ("k_futures", C.c_int32), ("n_agent", C.c_int32), ("n_pl", C.c_int32), ("n_tl", C.c_int32),
        ("n_step", C.c_int32),
        ("enable_check_collided", C.c_int32), ("enable_check_run_road_edge", C.c_int32),
        ("enable_check_run_red_light", C.c_int32), ("enable_check_passive", C.c_int32),
        ("check_state", c_f32p), ("check_valid", c_u8p), ("agent_type", c_i32p), ("agent_size", c_f32p),
        ("map_valid", c_u8p), ("map_type", c_i32p), ("map_pos", c_f32p), ("map_dir", c_f32p),
        ("tl_valid", c_u8p), ("tl_state", c_i32p), ("tl_pos", c_f32p),
        ("collided", c_u8p), ("collided_this_step", c_u8p), ("run_road_edge", c_u8p), ("run_road_edge_this_step", c_u8p),
        ("run_red_light", c_u8p), ("run_red_light_this_step", c_u8p), ("passive", c_u8p), ("passive_this_step", c_u8p),
        ("n_tl_step", C.c_int32), ("agent_goal", c_f32p), ("goal_reached", c_u8p), ("goal_reached_this_step", c_u8p),
    ]


class TbEncodeIO(C.Structure):
    _fields_ = [
        ("n_scene", C.c_int32), ("n_agent", C.c_int32), ("n_pl", C.c_int32), ("n_tl", C.c_int32), ("n_hist", C.c_int32),
        ("agent_valid", c_u8p), ("agent_pos", c_f32p), ("agent_yaw", c_f32p), ("agent_vel", c_f32p), ("agent_spd", c_f32p),
        ("agent_acc", c_f32p), ("agent_yaw_rate", c_f32p), ("agent_type", c_i32p), ("agent_size", c_f32p),
        ("map_valid", c_u8p), ("map_type", c_i32p), ("map_pos", c_f32p), ("map_dir", c_f32p),
        ("tl_valid", c_u8p), ("tl_state", c_i32p), ("tl_pos", c_f32p), ("tl_dir", c_f32p),
        ("map_feature", c_f32p), ("map_feature_valid", c_u8p), ("agent_feature", c_f32p), ("tl_feature", c_f32p),
        ("latent_mean", c_f32p), ("latent_valid", c_u8p), ("dest_logits", c_f32p),
        ("ext_agent_attr", c_f32p), ("ext_agent_pe", c_f32p), ("ext_map_attr", c_f32p), ("ext_map_pe", c_f32p),
        ("ext_tl_attr", c_f32p), ("ext_tl_pe", c_f32p),
    ]


class TbPostIO(C.Structure):
    _fields_ = [
        ("n_scene", C.c_int32), ("n_agent", C.c_int32), ("n_pred", C.c_int32), ("n_step", C.c_int32), ("d_traj", C.c_int32),
        ("k_pred", C.c_int32), ("score_temperature", C.c_float), ("n_mpa", C.c_int32), ("mpa_nms_thresh", C.c_float * 3),
        ("n_mtr", C.c_int32), ("mtr_nms_thresh", C.c_float * 3), ("use_ade", C.c_int32),
        ("valid", c_u8p), ("scores", c_f32p), ("trajs", c_f32p), ("agent_type", c_i32p),
        ("waymo_trajs", c_f32p), ("waymo_yaw_bbox", c_f32p), ("waymo_spd", c_f32p), ("waymo_scores", c_f32p),
        ("waymo_valid", c_u8p), ("mode_idx", c_i32p), ("traj_strides", C.c_int64 * 4),
    ]


class TbMetricIO(C.Structure):
    _fields_ = [
        ("n_scene", C.c_int32), ("n_agent", C.c_int32), ("k_futures", C.c_int32), ("n_step", C.c_int32),
        ("loss_for_teacher_forcing", C.c_int32),
        ("pred_valid", c_u8p), ("pred_states", c_f32p), ("override_masks", c_u8p), ("gt_valid", c_u8p), ("gt_states", c_f32p),
        ("agent_role", c_u8p), ("agent_type", c_i32p),
        ("outside_map", c_u8p), ("collided", c_u8p), ("run_road_edge", c_u8p), ("run_red_light", c_u8p), ("passive", c_u8p),
        ("goal_reached", c_u8p), ("dest_reached", c_u8p), ("out", C.POINTER(C.c_double)),
    ]


class TbPosteriorIO(C.Structure):
    _fields_ = [
        ("n_scene", C.c_int32), ("n_agent", C.c_int32), ("n_pl", C.c_int32), ("n_tl", C.c_int32), ("n_step", C.c_int32),
        ("agent_valid", c_u8p), ("agent_pos", c_f32p), ("agent_yaw", c_f32p), ("agent_vel", c_f32p), ("agent_spd", c_f32p),
        ("agent_acc", c_f32p), ("agent_yaw_rate", c_f32p), ("agent_type", c_i32p), ("agent_size", c_f32p),
        ("tl_valid", c_u8p), ("tl_state", c_i32p), ("tl_pos", c_f32p), ("tl_dir", c_f32p),
        ("map_feature", c_f32p), ("map_feature_valid", c_u8p), ("latent_mean", c_f32p), ("latent_valid", c_u8p),
    ]


class TbTrainIO(C.Structure):
    _fields_ = [
        ("n_scene", C.c_int32), ("n_agent", C.c_int32), ("n_step", C.c_int32), ("n_pl", C.c_int32),
        ("w_collision", C.c_float), ("reduce_collision_with_max", C.c_int32), ("use_il_loss", C.c_int32),
        ("crit_pos", C.c_int32), ("crit_rot", C.c_int32), ("angular_type", C.c_int32), ("crit_spd", C.c_int32),
        ("w_pos", C.c_float), ("w_rot", C.c_float), ("w_spd", C.c_float),
        ("use_vae_kl", C.c_int32), ("use_diffbar_reward", C.c_int32), ("use_goal", C.c_int32),
        ("kl_for_unseen_agent", C.c_int32), ("loss_for_teacher_forcing", C.c_int32), ("step_training_start", C.c_int32),
        ("kl_balance_scale", C.c_float), ("kl_free_nats", C.c_float),
        ("pred_valid", c_u8p), ("pred_states", c_f32p), ("override_masks", c_u8p), ("gt_valid", c_u8p), ("gt_states", c_f32p),
        ("agent_size", c_f32p), ("dest_logits", c_f32p), ("goal_valid", c_u8p), ("gt_dest", c_i32p),
        ("post_mean", c_f32p), ("post_valid", c_u8p), ("prior_mean", c_f32p), ("prior_valid", c_u8p),
        ("diffbar_rewards", c_f32p), ("diffbar_rewards_valid", c_u8p), ("out", C.POINTER(C.c_double)),
        ("relevant", c_u8p), ("irrelevant_draw", c_u8p),
    ]


EXPORTS = (
    "tb_create", "tb_destroy", "tb_last_error", "tb_version", "tb_load_weight", "tb_finalize_weights",
    "tb_rollout", "tb_rollout_begin", "tb_rollout_step", "tb_rollout_step_ex", "tb_check_status", "tb_rollout_state", "tb_encode_scene", "tb_set_timing", "tb_get_timing",
    "tb_rule_checks", "tb_post_process", "tb_metric_partials", "tb_struct_sizes", "tb_encode_posterior", "tb_train_partials",
    "tb_forward", "tb_graph_stats", "tb_latent_sample", "tb_dest_sample", "tb_precision_state", "tb_precision_note", "tb_precision_restore",
    "tb_host_onehot_index",
)

_lib: Optional[C.CDLL] = None


def lib_path() -> str:
    return _LIB_PATH


VALIDATED_RUNTIME_HIP = ("7.0",)  # major.minor of the in-process HIP runtimes (torch.version.hip) the GPU suite has run on


def build_info() -> Dict:
    """trafficbots_amd/lib/build_info.json (written by __graft_entry__.build next to the library): the fingerprint of the sources, the
    compiler + flag set it was built with and whether that pair is a validated one (trafficbots_amd/csrc/toolchain.json), the result
    of the ISA hazard lint.  {} for a library built some other way (the Makefile)."""
    import json

    path = os.path.join(os.path.dirname(_LIB_PATH), "build_info.json")
    try:
        with open(path) as f:
            return json.load(f)
    except Exception:
        return {}


def _warn_if_unvalidated() -> None:
    """The library is at the edge of what the compiler schedules correctly (profiles/r05_experiments.txt item 24): one built with a
    compiler release / flag set the GPU suite has not seen, or loaded under another HIP runtime generation, says so -- once per
    process, loudly; TB_REQUIRE_VALIDATED_TOOLCHAIN=1 turns the warning into an error."""
    import warnings

    problems = []
    info = build_info()
    tc = info.get("toolchain")
    if tc is not None and not tc.get("validated", False):
        problems.append(f"built with an unvalidated toolchain ({tc.get('why')})")
    try:
        import torch

        rt = getattr(torch.version, "hip", None)
        if rt and ".".join(rt.split(".")[:2]) not in VALIDATED_RUNTIME_HIP:
            problems.append(f"running under HIP runtime {rt}; validated: {', '.join(VALIDATED_RUNTIME_HIP)}.x")
    except Exception:
        pass
    if not problems:
        return
    msg = ("trafficbots_amd: " + "; ".join(problems) + " -- results are unverified on this combination: run `pytest -m gpu` and "
           "tests/probes/gpu_guard_pages.py before trusting them (trafficbots_amd/csrc/toolchain.json)")
    if os.environ.get("TB_REQUIRE_VALIDATED_TOOLCHAIN") == "1":
        raise RuntimeError(msg)
    warnings.warn(msg, RuntimeWarning, stacklevel=3)


def load() -> C.CDLL:
    """dlopen the HIP library (raises if it has not been built: run `python -c 'import
    __graft_entry__ as g; g.build()'` or `make -C trafficbots_amd/csrc`)."""
    global _lib
    if _lib is not None:
        return _lib
    # torch's own HIP runtime must be in the process BEFORE this library is dlopened: the wheel bundles its libamdhip64, and a process
    # that loads the system's copy first (this library's DT_NEEDED) and torch's afterwards ends up with two runtimes of which the second
    # sees no device ("no HIP device visible" from tb_create after `build(); smoke()` in one interpreter, round 4).  With torch first the
    # loader resolves this library's libamdhip64 to the one already loaded.
    import torch  # noqa: F401
    if not os.path.exists(_LIB_PATH):
        raise RuntimeError(
            f"trafficbots_amd: HIP library not built ({_LIB_PATH} missing). There is no CPU fallback; "
            "build it with __graft_entry__.build()."
        )
    _warn_if_unvalidated()
    lib = C.CDLL(_LIB_PATH)
    lib.tb_create.argtypes = [C.POINTER(TbConfig), C.POINTER(C.c_void_p)]
    lib.tb_create.restype = C.c_int
    lib.tb_destroy.argtypes = [C.c_void_p]
    lib.tb_destroy.restype = None
    lib.tb_last_error.argtypes = [C.c_void_p]
    lib.tb_last_error.restype = C.c_char_p
    lib.tb_version.argtypes = []
    lib.tb_version.restype = C.c_char_p
    lib.tb_load_weight.argtypes = [C.c_void_p, C.c_char_p, c_f32p, C.c_int64]
    lib.tb_load_weight.restype = C.c_int
    lib.tb_finalize_weights.argtypes = [C.c_void_p, C.c_void_p]
    lib.tb_finalize_weights.restype = C.c_int
    lib.tb_rollout.argtypes = [C.c_void_p, C.POINTER(TbRolloutIO), C.c_void_p]
    lib.tb_rollout.restype = C.c_int
    lib.tb_rollout_begin.argtypes = [C.c_void_p, C.POINTER(TbRolloutIO), C.c_void_p]
    lib.tb_rollout_begin.restype = C.c_int
    lib.tb_rollout_step.argtypes = [C.c_void_p, C.c_void_p]
    lib.tb_rollout_step.restype = C.c_int
    lib.tb_rollout_step_ex.argtypes = [C.c_void_p, C.POINTER(TbStepOverride), C.c_void_p]
    lib.tb_rollout_step_ex.restype = C.c_int
    lib.tb_check_status.argtypes = [C.c_void_p, C.c_void_p]
    lib.tb_check_status.restype = C.c_int
    lib.tb_rollout_state.argtypes = [C.c_void_p, c_f32p, c_u8p, c_f32p, C.c_void_p]
    lib.tb_rollout_state.restype = C.c_int
    lib.tb_encode_scene.argtypes = [C.c_void_p, C.POINTER(TbEncodeIO), C.c_void_p]
    lib.tb_encode_scene.restype = C.c_int
    lib.tb_rule_checks.argtypes = [C.c_void_p, C.POINTER(TbRuleIO), C.c_void_p]
    lib.tb_rule_checks.restype = C.c_int
    lib.tb_post_process.argtypes = [C.c_void_p, C.POINTER(TbPostIO), C.c_void_p]
    lib.tb_post_process.restype = C.c_int
    lib.tb_metric_partials.argtypes = [C.c_void_p, C.POINTER(TbMetricIO), C.c_void_p]
    lib.tb_metric_partials.restype = C.c_int
    lib.tb_encode_posterior.argtypes = [C.c_void_p, C.POINTER(TbPosteriorIO), C.c_void_p]
    lib.tb_encode_posterior.restype = C.c_int
    lib.tb_train_partials.argtypes = [C.c_void_p, C.POINTER(TbTrainIO), C.c_void_p]
    lib.tb_train_partials.restype = C.c_int
    lib.tb_struct_sizes.argtypes = [C.POINTER(C.c_int32)]
    lib.tb_struct_sizes.restype = None
    lib.tb_latent_sample.argtypes = [C.c_void_p, C.POINTER(TbLatentSampleIO), C.c_void_p]
    lib.tb_latent_sample.restype = C.c_int
    lib.tb_dest_sample.argtypes = [C.c_void_p, C.POINTER(TbDestSampleIO), C.c_void_p]
    lib.tb_dest_sample.restype = C.c_int
    sizes = (C.c_int32 * 13)()
    lib.tb_struct_sizes(sizes)
    mine = [C.sizeof(x) for x in (TbConfig, TbRolloutIO, TbEncodeIO, TbRuleIO, TbPostIO, TbMetricIO)] + [C.sizeof(C.c_void_p)]
    mine += [C.sizeof(TbPosteriorIO), C.sizeof(TbTrainIO), C.sizeof(TbStepOverride), C.sizeof(TbForwardIO)]
    mine += [C.sizeof(TbLatentSampleIO), C.sizeof(TbDestSampleIO)]
    lib.tb_forward.argtypes = [C.c_void_p, C.POINTER(TbForwardIO), C.c_void_p]
    lib.tb_forward.restype = C.c_int
    if list(sizes) != mine:
        raise RuntimeError(f"trafficbots_amd: ctypes struct layouts {mine} do not match the library's {list(sizes)} (stale build?)")
    lib.tb_precision_state.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
    lib.tb_precision_state.restype = C.c_int
    lib.tb_precision_note.argtypes = [C.c_void_p]
    lib.tb_precision_note.restype = C.c_char_p
    lib.tb_precision_restore.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
    lib.tb_precision_restore.restype = C.c_int
    lib.tb_host_onehot_index.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p]
    lib.tb_host_onehot_index.restype = None
    lib.tb_set_timing.argtypes = [C.c_void_p, C.c_int]
    lib.tb_set_timing.restype = C.c_int
    lib.tb_get_timing.argtypes = [C.c_void_p, c_f32p]
    lib.tb_get_timing.restype = C.c_int
    _lib = lib
    return lib


# Test hook (tests/probes/gpu_guard_pages.py): an object with `shadow(tensor) -> int` and `writeback()`.  When set, every tensor handed
# to the C ABI is replaced by a copy in a buffer with unmapped memory on both sides (tests/guard/tb_guard.cpp), and copied back after
# the call -- an out-of-bounds access of a kernel becomes a GPU memory fault.  None in every product run.
guard_hook = None


def ptr(t, ctype):
    """Device pointer of a contiguous torch tensor (or NULL for None) as a ctypes pointer."""
    if t is None:
        return ctype()  # NULL
    assert t.is_contiguous(), "tensor handed to the C ABI must be contiguous"
    if guard_hook is not None and t.is_cuda:
        return C.cast(guard_hook.shadow(t), ctype)
    return C.cast(t.data_ptr(), ctype)


def make_config(cfg: Dict) -> TbConfig:
    c = TbConfig()
    c.time_step_current = int(cfg["time_step_current"])
    c.time_step_sim_start = int(cfg["time_step_sim_start"])
    c.dt = 0.1  # Dynamics default dt (src/utils/dynamics.py:13)
    dyn = cfg["dynamics"]
    for i, k in enumerate(("veh", "ped", "cyc")):  # type order of `Dynamics.agent_dynamics` (dynamics.py:23-27)
        c.max_acc[i] = float(dyn[k]["max_acc"])
        c.max_yaw_rate[i] = float(dyn[k]["max_yaw_rate"])
    c.action_log_std = float(cfg["action_head"]["log_std"])
    c.latent_log_std = float(cfg["model"]["latent_encoder"]["latent_prior"]["log_std"])
    # not a key of the reference: "fp32" (default: fp16-pair XDL kernels, automatic fallback to the exact kernels when a tensor or an
    # activation leaves their range), "bf16" (BASELINE configs 4/5), "fp32_exact" (fp32 MFMA kernels from the start)
    prec = str(cfg.get("operand_precision", "fp32"))
    if prec not in ("fp32", "bf16", "fp32_exact"):
        raise ValueError(f"operand_precision must be 'fp32', 'bf16' or 'fp32_exact', got {prec!r}")
    c.operand_precision = {"fp32": 0, "bf16": 1, "fp32_exact": 2}[prec]
    # not keys of the reference either: `library_switches: {step_l2_warmers: 1, ...}` -> tb_config.sw (include/trafficbots_hip.h;
    # every field 0 = automatic, the default)
    for name, value in dict(cfg.get("library_switches") or {}).items():
        if name not in SWITCH_FIELDS:
            raise ValueError(f"library_switches: unknown switch {name!r} (known: {', '.join(SWITCH_FIELDS)})")
        setattr(c.sw, name, int(value))
    return c
