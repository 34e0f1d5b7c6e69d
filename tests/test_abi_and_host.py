"""CPU-side tests: the C-ABI library loads and exports every symbol include/*.h declares (no compute
without a GPU), config / synthetic-data / host-logic units."""
import ctypes
import json
import os
import re
import sys

import numpy as np
import pytest
import torch

from conftest import GOLDEN_DIR, ROOT


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as ge

    ge.build()
    from trafficbots_amd import hip

    header = open(os.path.join(ROOT, "include", "trafficbots_hip.h")).read()
    declared = set(re.findall(r"\b(tb_[a-z_]+)\s*\(", header))
    declared -= {"tb_ctx", "tb_stream"}
    assert declared, "no declarations parsed"
    lib = ctypes.CDLL(hip.lib_path())
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in the header but not exported"
    assert set(hip.EXPORTS) == declared
    lib.tb_version.restype = ctypes.c_char_p
    assert b"gfx950" in lib.tb_version()


def test_struct_layouts_match_header_field_order():
    from trafficbots_amd import hip

    header = open(os.path.join(ROOT, "include", "trafficbots_hip.h")).read()
    for cname, cls in (("tb_rollout_io", hip.TbRolloutIO), ("tb_encode_io", hip.TbEncodeIO), ("tb_config", hip.TbConfig),
                       ("tb_latent_sample_io", hip.TbLatentSampleIO), ("tb_dest_sample_io", hip.TbDestSampleIO)):
        body = header[header.index(f"typedef struct {cname} {{"):header.index(f"}} {cname};")]
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        names = []
        for decl in body.split("{", 1)[1].split(";"):
            decl = decl.strip()
            if not decl:
                continue
            for part in decl.split(","):
                m = re.search(r"([A-Za-z_][A-Za-z0-9_]*)\s*(\[\d+\])?$", part.strip())
                names.append(m.group(1))
        assert names == [f[0] for f in cls._fields_], cname


def test_no_gpu_means_loud_failure():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from trafficbots_amd.runtime import HipEngine
    from trafficbots_amd.config import load_model_config

    with pytest.raises(RuntimeError):
        HipEngine(load_model_config())


def test_config_resolver_and_unsupported_branches():
    from trafficbots_amd.config import DEFAULT_MODEL_CONFIG, load_model_config, resolve_interpolations

    cfg = {"hidden_dim": 128, "model": {"hidden_dim": "${..hidden_dim}", "tf_cfg": {"d_model": "${...hidden_dim}"},
                                        "a": {"x": 1, "y": "${.x}"}, "b": "${.a}"}}
    r = resolve_interpolations(cfg)
    assert r["model"]["hidden_dim"] == 128 and r["model"]["tf_cfg"]["d_model"] == 128
    assert r["model"]["a"]["y"] == 1 and r["model"]["b"] == {"x": 1, "y": 1}
    with pytest.raises(ValueError):
        resolve_interpolations({"a": "${.b}", "b": "${.a}"})
    assert load_model_config()["model"]["tf_cfg"]["n_head"] == DEFAULT_MODEL_CONFIG["model"]["tf_cfg"]["n_head"]
    for key, val in (("model.goal_manager.goal_attr_mode", "goal_xy"), ("model.resample_latent", True),
                     ("model.interaction_first", False)):
        with pytest.raises(NotImplementedError):
            load_model_config(overrides={key: val})
    # the flag-gated traffic-rule checks are built (tb_rule_checks): accepted
    assert load_model_config(overrides={"traffic_rule_checker.enable_check_collided": True})["traffic_rule_checker"]["enable_check_collided"]


def test_state_dict_spec_matches_reference_keys():
    from trafficbots_amd import synth

    ref = json.load(open(os.path.join(GOLDEN_DIR, "state_dict_keys.json")))
    spec = synth.state_dict_spec()
    assert set(spec.keys()) == set(ref.keys())
    for k, shp in spec.items():
        assert list(shp) == ref[k], k
    sd = synth.make_state_dict(7)
    sd2 = synth.make_state_dict(7)
    assert all((sd[k] == sd2[k]).all() for k in sd)
    # aliases share values (shared_transformer_as=True)
    a = "model.latent_encoder.transformer_as2pl.layers.0.linear1.weight"
    assert (sd[a] == sd["model.transformer_as2pl.layers.0.linear1.weight"]).all()
    assert abs(float(sd["action_head.log_std.0"][0]) + 2.0) < 1e-9


def test_synthetic_scene_is_seed_stable():
    from trafficbots_amd import synth

    a = synth.make_batch(123, 2, n_agent=6, n_pl=9, n_tl=5, p_invalid_agent=0.3, p_late_spawn=0.5)
    b = synth.make_batch(123, 2, n_agent=6, n_pl=9, n_tl=5, p_invalid_agent=0.3, p_late_spawn=0.5)
    assert all((a[k] == b[k]).all() for k in a)
    assert a["history/agent/valid"].shape == (2, 11, 6) and a["map/pos"].shape == (2, 9, 20, 2)
    assert a["history/agent/type"].sum(-1).min() == 1
    # shard consistency: rank shards of a global batch are slices of it
    g = synth.make_batch(50, 4, n_agent=4, n_pl=8)
    s = synth.make_batch(50, 2, scene_offset=2, n_agent=4, n_pl=8)
    assert (g["map/pos"][2:] == s["map/pos"]).all()
    # pinned raw-stream values (guards against numpy changing PCG64 / our float mapping)
    u = synth.RawStream(7).u01((3,))
    assert np.allclose(u, [0.625095466605, 0.89721380097, 0.775685690245], atol=1e-9), u


def test_teacher_forcing_mask_matches_oracle():
    from oracle.trafficbots_oracle import Oracle
    from trafficbots_amd.runtime import teacher_forcing_mask

    g = torch.Generator().manual_seed(0)
    valid = torch.rand(5, 11, 7, generator=g) > 0.4
    o = Oracle.__new__(Oracle)
    for cfg in ({"step_spawn_agent": 10, "step_warm_start": 10}, {"step_spawn_agent": 3, "step_warm_start": 1},
                {"step_spawn_agent": 0, "step_warm_start": -1}):
        assert (teacher_forcing_mask(valid.clone(), **cfg) == Oracle.teacher_forcing_mask(o, valid.clone(), cfg)).all()


def test_distributions_interface_and_no_host_fallback():
    """The distribution mirrors keep the reference's surface (`mean`, `valid`, `repeat_interleave_`, `sample`, `log_prob`) but their
    arithmetic lives in the HIP library (`tb_latent_sample`, `tb_dest_sample`; tests/test_gpu_boundary.py checks it against
    torch.distributions): an object that is not bound to a device engine raises instead of computing on the host."""
    from trafficbots_amd.distributions import DestCategorical, DiagGaussian

    g = torch.Generator().manual_seed(1)
    mean, log_std = torch.randn(4, 5, 16, generator=g), torch.full((16,), -1.0)
    valid = torch.rand(4, 5, generator=g) > 0.3
    d = DiagGaussian(mean, log_std, valid=valid)
    d.repeat_interleave_(3, 0)
    assert d.mean.shape == (12, 5, 16) and torch.equal(d.mean[3:6], mean[1:2].expand(3, -1, -1)) and torch.equal(d.valid, valid.repeat_interleave(3, 0))
    eps, det = d._draws(torch.zeros(12, 5, dtype=torch.bool), None, torch.Generator().manual_seed(2))
    assert eps.shape == (12, 5, 16) and det.shape == (12, 5)
    assert d._draws(True, None, None) == (None, None)
    with pytest.raises(RuntimeError, match="no host fallback"):
        d.sample(True)
    with pytest.raises(RuntimeError, match="no host fallback"):
        d.log_prob(mean.repeat_interleave(3, 0))
    c = DestCategorical(logits=torch.randn(3, 4, 9, generator=g), valid=torch.ones(3, 4, dtype=torch.bool))
    c.repeat_interleave_(2, 0)
    assert c.valid.shape == (6, 4) and c._from_probs
    for call in (lambda: c.sample(True), lambda: c.log_prob(torch.zeros(6, 4, dtype=torch.long)), lambda: c.probs):
        with pytest.raises(RuntimeError, match="no host fallback"):
            call()


def test_rollout_buffer_flatten_repeat():
    from trafficbots_amd.waymo_motion import RolloutBuffer

    b = RolloutBuffer(1, 90, 10)
    assert b.step_future_start == 10
    n, a, s, k = 6, 3, 4, 3
    b.valid = torch.arange(n * a * s).view(n, a, s) % 2 == 0
    b.override_masks = b.valid.clone()
    b.preds = torch.arange(n * a * s * 4, dtype=torch.float32).view(n, a, s, 4)
    b.violations = {"outside_map": b.valid.clone()}
    b.latent_log_probs = torch.zeros(n, a, s)
    b.action_log_probs = torch.zeros(n, a, s)
    p0 = b.preds.clone()
    b.flatten_repeat(k)
    assert b.preds.shape == (2, a, k, s, 4)
    assert torch.equal(b.preds[1, 2, 1], p0[1 * k + 1, 2])


def test_training_split_batch_has_no_history_keys():
    """A reference TRAINING batch (`tensor_size_train`, `data_h5_womd.py:85-117`) carries no "history/*" keys: the scene is the
    first n_hist steps of "agent/*" / "tl_stop/*" (`scene_centric.py:92-121`, prefix "" in training).  Same device-layout scene
    as from a validation batch whose history equals those steps; `warm_ok` follows n_hist."""
    from trafficbots_amd import synth
    from trafficbots_amd.runtime import gt_from_batch, scene_from_batch

    val = synth.make_val_batch(31, 2, n_agent=9, n_pl=20, n_tl=6)
    train = {k: v for k, v in val.items() if not k.startswith("history/")}
    a = scene_from_batch(val, "cpu", 11)
    b = scene_from_batch(train, "cpu", 11)
    assert set(a) == set(b)
    for k in a:
        if not torch.is_tensor(a[k]):  # "warm_ok", "_tf_params": host-side facts
            assert a[k] == b[k]
        else:
            assert torch.equal(a[k], b[k]), k
    with pytest.raises(KeyError):
        scene_from_batch({k: v for k, v in train.items() if not k.startswith("agent/")}, "cpu", 11)
    # an exit at step 12 is inside a 14-step history but not inside the default 11
    v = np.ones_like(np.asarray(train["agent/valid"]))
    v[:, 12:, 0] = False
    t3 = dict(train, **{"agent/valid": v})
    assert gt_from_batch(t3, "cpu", 11)["warm_ok"] is True and gt_from_batch(t3, "cpu", 14)["warm_ok"] is False


def _yaml_shaped_config():
    """A dict with the keys and nesting of `configs/model/traffic_bots.yaml` as Hydra would hand it to `instantiate` (with the
    relative interpolations still in place), built from this package's default tree -- the reference's file itself does not travel."""
    import copy

    from trafficbots_amd.config import DEFAULT_MODEL_CONFIG

    cfg = copy.deepcopy(DEFAULT_MODEL_CONFIG)
    cfg["_target_"] = "pl_modules.waymo_motion.WaymoMotion"
    cfg["model"]["_target_"] = "models.traffic_bots.TrafficBots"
    cfg["model"]["hidden_dim"] = "${..hidden_dim}"
    cfg["model"]["tf_cfg"]["d_model"] = "${...hidden_dim}"
    cfg["pre_processing"] = {
        "scene_centric": {"_target_": "data_modules.scene_centric.SceneCentricPreProcessing"},
        "input": dict(cfg["pre_processing"]["input"], _target_="data_modules.sc_input.SceneCentricInput",
                      pose_pe={"map": "pe_xy_yaw", "tl": "${.map}", "agent": "${.map}"}),
        "latent": {"_target_": "data_modules.sc_latent.SceneCentricLatent", "pe_dim": "${..input.pe_dim}", "pose_pe": "${..input.pose_pe}",
                   "perturb_input_to_latent": False, "dropout_p_history": -1, "max_meter": 50.0, "max_rad": 3.14},
    }
    cfg.update(n_video_batch=3, interactive_challenge=False, step_detach_hidden=-1, p_drop_hidden=-1.0, lr_goal=1e-3,
               optimizer={"_target_": "torch.optim.AdamW", "lr": 3e-4}, lr_scheduler=None,
               sub_womd_reactive_replay={"activate": False}, sub_womd_joint_future_pred={"activate": False},
               data_size={"agent/valid": [91, 64], "map/valid": [1024, 20]}, wb_artifact=None)
    return cfg


def test_hydra_style_instantiation_of_the_mirror():
    """`trafficbots_amd.instantiate(cfg)` = `hydra.utils.instantiate(cfg.model)` of `src/run.py:34-36`: the `_target_` of the
    reference's task module resolves to the mirror and the constructor takes the reference's keyword arguments
    (`waymo_motion.py:28-62`).  Without a GPU the call must get as far as creating the engine (configuration accepted) and fail
    THERE, loudly; unsupported ablation branches and foreign targets are rejected before that."""
    import copy

    import trafficbots_amd
    from trafficbots_amd.config import config_from_hydra_kwargs

    cfg = _yaml_shaped_config()
    resolved = config_from_hydra_kwargs({k: v for k, v in copy.deepcopy(cfg).items() if k != "_target_"})
    assert resolved["model"]["tf_cfg"]["d_model"] == 128 and resolved["pre_processing"]["input"]["pose_pe"]["agent"] == "pe_xy_yaw"
    assert resolved["optimizer"]["lr"] == 3e-4 and resolved["data_size"]["agent/valid"] == [91, 64]  # carried along, unused
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="no HIP device"):
            trafficbots_amd.instantiate(copy.deepcopy(cfg))
    bad = copy.deepcopy(cfg)
    bad["model"]["interaction_first"] = False
    with pytest.raises(NotImplementedError, match="interaction_first"):
        trafficbots_amd.instantiate(bad)
    with pytest.raises(NotImplementedError, match="_target_"):
        trafficbots_amd.instantiate(dict(copy.deepcopy(cfg), _target_="pl_modules.something_else.Model"))
    missing = copy.deepcopy(cfg)
    del missing["dynamics"]
    with pytest.raises(KeyError):
        trafficbots_amd.instantiate(missing)
    ref_yaml = "/root/reference/configs/model/traffic_bots.yaml"
    if os.path.exists(ref_yaml):  # build container only: the reference's own file passes as it is
        import yaml

        y = yaml.safe_load(open(ref_yaml))
        got = config_from_hydra_kwargs(dict(y, data_size={}))
        assert got["teacher_forcing_reactive_replay"]["step_spawn_agent"] == 90 and got["n_joint_future"] == 6


def test_pre_processed_scene_carries_the_reference_keys():
    """the twelve "input/*" arguments of `encode_input_features` (+ the aliased "latent_prior/*", + "ref/*") are on the
    pre-processed scene; attr / pe are zero-storage stand-ins that point back at the scene"""
    from trafficbots_amd import synth
    from trafficbots_amd.runtime import scene_from_batch
    from trafficbots_amd.waymo_motion import _with_reference_keys

    scene = _with_reference_keys(scene_from_batch(synth.make_batch(5, 2, n_agent=7, n_pl=9, n_tl=3), "cpu", 11))
    names = ("agent_valid", "agent_attr", "agent_pe", "agent_pos", "map_valid", "map_attr", "map_pe", "map_pos", "tl_valid", "tl_attr",
             "tl_pe", "tl_pos")
    input_dict = {k.split("input/")[-1]: v for k, v in scene.items() if "input/" in k}  # the reference harness' idiom (:907)
    assert set(input_dict) == set(names)
    shapes = {"agent_attr": (2, 11, 7, 11), "agent_pe": (2, 11, 7, 96), "map_attr": (2, 9, 20, 31), "map_pe": (2, 9, 20, 96),
              "tl_attr": (2, 11, 3, 5), "tl_pe": (2, 11, 3, 96), "map_pos": (2, 9, 2), "agent_valid": (2, 11, 7)}
    for k, shp in shapes.items():
        assert tuple(input_dict[k].shape) == shp, k
    assert input_dict["agent_valid"].dtype == torch.bool and input_dict["agent_valid"].any(1).shape == (2, 7)
    for k in ("agent_attr", "agent_pe", "map_attr", "map_pe", "tl_attr", "tl_pe"):
        # (a WEAK reference since round 6: scene -> stand-in -> scene was a cycle that kept each batch's device slab alive until the
        # cyclic collector ran)
        assert input_dict[k]._tb_scene() is scene and input_dict[k].untyped_storage().nbytes() == 4
        assert scene["latent_prior/" + k] is input_dict[k]
    assert scene["ref/agent_type"].shape == (2, 7, 3) and scene["ref/agent_type"].sum(-1).max() == 1
    assert scene["ref/map_type"].shape == (2, 9, 11) and scene["ref/agent_state"].shape == (2, 11, 7, 4)


def test_bench_accounting_functions():
    """VERDICT r05 task 3: the bench line's own arithmetic.  (1) the per-CU load path counts the workgroups a CU processes during a
    launch -- K = 6 (768 tiles, three per CU) at r05's 129.1 us must read ~0.44 of the L1 fill peak and bind as "load-path", not 0.145
    and "latency"; (2) `what_binds` names a resource only from 30 % of its peak; (3) the encoder fraction is priced on the pipe that
    executes (hoisted flops x 3 MFMAs over 2.5 PFLOP/s) and stays below 1, the SURVEY figure is kept as `frac_nominal`-style input;
    (4) the floor takes its cycle counts from a stage profile record, not from constants in the code."""
    import bench

    one = bench.load_path(129.12, 64, 256, 64.0, 2)
    k6 = bench.load_path(129.12, 64, 256, 64.0, 2, n_workgroup=768)
    assert abs(k6["achieved_B_per_clk_per_CU"] - 3.0 * one["achieved_B_per_clk_per_CU"]) < 1e-9 and k6["workgroups_per_cu"] == 3.0
    assert 0.40 < k6["frac"] < 0.48, k6["frac"]
    assert abs(k6["frac_of_sustained"] - k6["frac"] * 64.0 / 42.0) < 1e-12
    head = bench.load_path(86.0, 64, 256, 64.0, 4, n_workgroup=128)  # 128 tiles: one workgroup per ACTIVE CU
    assert head["workgroups_per_cu"] == 1.0 and head["frac"] < 0.5
    assert bench.what_binds(0.041, 0.16, 0.105, k6["frac"]) == "load-path"
    assert bench.what_binds(0.128, 0.13, 0.076, head["frac"]) in ("latency/load-path", "load-path")
    assert bench.what_binds(0.05, 0.10, 0.05, 0.20) == "latency/load-path"
    assert bench.what_binds(0.60, 0.10, 0.60, 0.20) == "mfma" and bench.what_binds(None, 0.70, 0.10, 0.20) == "hbm"
    ex, nominal = bench.flops_encode_executed(64, 256), bench.flops_encode(64, 256)
    assert ex < 3.0 * nominal  # hoisted: less than three times the un-hoisted count
    frac = ex * 32 / (1.22e-3) / 1e12 / bench.PEAK_BF16_MFMA_TFLOPS
    assert 0.05 < frac < 1.0, frac
    sc = bench.stage_constants()
    assert set(sc) >= {"cold_cycles", "attention_walks", "layernorms_21", "source"} and sc["cold_cycles"] > 0
    fl = bench.structural_floor(86.0, 64, 256, 64.0, 4, flops=176.1e6 * 32)
    assert abs(fl["cold_start_us"] - sc["cold_cycles"] / bench.SHADER_CLK * 1e6) < 1e-9
    assert fl["floor_us"] < 86.0 and 0.0 < fl["frac_of_nominal_roof_at_floor"] < 1.0


def test_isa_hazard_lint_finds_the_planted_hazards_and_passes_the_built_objects():
    """VERDICT r05 task 2: the machine-code canary.  tools/isa_waw_lint.py walks a kernel's instructions along its control-flow graph
    with the queue of outstanding vector-memory operations; an instruction that writes (or reads) a register an outstanding load will
    still write is the bug class that made the default-scheduler build of round 5 fault (a wait count the compiler did not insert).
    (1) On hand-written fragments (tools/microtests/waw_case.s) it reports exactly the planted hazards -- the round-5 pattern, a read
    behind a partial wait, a hazard behind a branch -- and none in the fixed forms (a full wait, a partial wait that covers the
    register, a load re-using a load's destination: loads return in order).  (2) The objects of the library as built here are clean."""
    import glob

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import isa_waw_lint as lint

    k = lint.parse_asm_text(open(os.path.join(ROOT, "tools", "microtests", "waw_case.s")).read().splitlines())
    got = {name: lint.lint_kernel(ins) for name, ins in k.items()}
    # rule 2: register-allocator traffic in front of a join block's exec restore -- the cause of the default-scheduler build's GPU
    # fault (profiles/r06_experiments.txt item 7): flagged in the planted bad form, clean in the good one and in every rule-1 case
    got2 = {name: lint.lint_exec_prologue(ins) for name, ins in k.items()}
    assert len(got2.pop("exec_prologue_bad")) == 1 and all(v == [] for v in got2.values()), got2
    assert got.pop("exec_prologue_bad") == [] and got.pop("exec_prologue_good") == []
    assert set(got) == {"bad", "good", "partial_wait_ok", "partial_wait_bad", "loads_only", "across_blocks"}
    assert len(got["bad"]) == 2 and "WRITES v108" in got["bad"][0] and "READS v108" in got["bad"][1]
    assert got["good"] == [] and got["partial_wait_ok"] == [] and got["loads_only"] == []
    assert len(got["partial_wait_bad"]) == 1 and "READS v5" in got["partial_wait_bad"][0]
    assert len(got["across_blocks"]) == 1 and "WRITES v3" in got["across_blocks"][0]
    objs = sorted(glob.glob(os.path.join(ROOT, "trafficbots_amd", "csrc", "build", "tb_step*.o")))
    if not objs or not os.path.exists(os.path.join(lint.LLVM, "llvm-objdump")):
        pytest.skip("no built objects / no llvm-objdump on this box")
    n_kernel, n_bad, report = lint.lint_paths(objs)
    assert n_kernel >= 10 and n_bad == 0, report[:10]


def test_toolchain_gate_refuses_unvalidated_compilers_and_flags(monkeypatch):
    """The flag set has ONE source (trafficbots_amd/csrc/toolchain.json: __graft_entry__.FLAGS, the Makefile and tools/build_variant.sh
    read it) and `build()` refuses a compiler release / flag set the GPU suite has not been run on; a library built anyway
    (TB_ALLOW_UNVALIDATED_TOOLCHAIN=1) warns when it is loaded, or fails with TB_REQUIRE_VALIDATED_TOOLCHAIN=1."""
    import json
    import warnings

    import __graft_entry__ as ge
    from trafficbots_amd import hip

    tc = json.load(open(os.path.join(ROOT, "trafficbots_amd", "csrc", "toolchain.json")))
    assert ge.FLAGS == tc["flags"] and "-amdgpu-sched-strategy=max-ilp" in ge.FLAGS
    mk = open(os.path.join(ROOT, "trafficbots_amd", "csrc", "Makefile")).read()
    assert "toolchain.json" in mk and "max-ilp" not in mk  # (no second copy of the flags)
    good = {"hip": tc["validated"][0]["hip"], "clang": tc["validated"][0]["clang"]}
    assert ge.toolchain_status(good)["validated"]
    other = dict(good, clang="AMD clang version 23.0.0git (roc-7.3.0)")
    st = ge.toolchain_status(other)
    assert not st["validated"] and "23.0.0" in st["why"]
    monkeypatch.setattr(ge, "FLAGS", [f for f in ge.FLAGS if "sched-strategy" not in f])
    assert not ge.toolchain_status(good)["validated"]
    monkeypatch.undo()
    # build() stops before compiling anything
    monkeypatch.setattr(ge, "compiler_identity", lambda hipcc=None: other)
    monkeypatch.setattr(ge, "_stale", lambda: True)
    monkeypatch.delenv("TB_ALLOW_UNVALIDATED_TOOLCHAIN", raising=False)
    with pytest.raises(RuntimeError, match="unvalidated toolchain"):
        ge.build()
    monkeypatch.undo()
    # load-time notice of a library built anyway
    monkeypatch.setattr(hip, "build_info", lambda: {"toolchain": {"validated": False, "why": "compiler X is not among the validated releases"}})
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        hip._warn_if_unvalidated()
    assert any("unvalidated toolchain" in str(x.message) for x in w)
    monkeypatch.setenv("TB_REQUIRE_VALIDATED_TOOLCHAIN", "1")
    with pytest.raises(RuntimeError, match="unvalidated toolchain"):
        hip._warn_if_unvalidated()


def test_pre_processed_scene_is_not_cyclic_garbage():
    """Round 6 (tests/probes/gpu_soak.py): the stand-ins' back-reference to their scene is weak -- dropping the last reference to a
    pre-processed scene frees it (and the device slab its tensors are views of) at once, by reference counting; the stand-ins of a
    scene that is gone raise instead of being taken for caller-made attributes."""
    import gc
    import weakref

    from trafficbots_amd import synth
    from trafficbots_amd.runtime import scene_from_batch
    from trafficbots_amd.waymo_motion import _scene_of_stand_ins, _with_reference_keys, retarget_stand_ins

    gc.collect()
    gc.disable()
    try:
        scene = _with_reference_keys(scene_from_batch(synth.make_batch(5, 2, n_agent=7, n_pl=9, n_tl=3), "cpu", 11))
        probe = weakref.ref(scene)
        attr = scene["input/agent_attr"]
        assert _scene_of_stand_ins((attr,)) is scene and _scene_of_stand_ins((torch.zeros(2),)) is None
        moved = type(scene)(scene)  # what staging.StagedBatch(scene) does: the entries live in another dict object now
        retarget_stand_ins(moved)
        del scene
        assert probe() is None, "the scene survived its last reference: a cycle"
        assert _scene_of_stand_ins((attr,)) is moved
        del moved
        with pytest.raises(RuntimeError, match="no longer exists"):
            _scene_of_stand_ins((attr,))
    finally:
        gc.enable()


def test_warm_schedule_is_generated_from_the_committed_stage_profile():
    """VERDICT r05 task 6: the L2 warmers' timetable is not a set of hand-copied cycle counts any more -- `tools/gen_warm_schedule.py`
    derives it from `profiles/stage_constants.json` (the stage profile `tools/gpu_stage_profile.py` measures on the GPU) into
    `csrc/tb_warm_schedule.inc`, which `tb_api.hip` includes.  The committed .inc must be what the script makes of the committed
    profile, the constants must be plausible (positive, ordered), and the source must not carry literal cycle counts beside them."""
    import json

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_warm_schedule as g

    sc = json.load(open(g.SRC))
    assert open(g.DST).read() == g.render(sc), "stale: run tools/gen_warm_schedule.py"
    s = g.schedule(sc)
    assert all(v > 0 for v in s.values())
    assert s["LAYER_W2_BEFORE_END"] < s["LAYER_W1_BEFORE_END"] < s["PL_BASE"] + s["PL_PER_BLOCK"] * (sc["shape"]["P"] // 32)
    assert abs(s["INTER_BASE"] + s["INTER_PER_BLOCK"] * (sc["shape"]["A"] // 32) - sc["c_half"]["C: interaction x3"] / 3) < 1e-6
    api = open(os.path.join(ROOT, "trafficbots_amd", "csrc", "tb_api.hip")).read()
    a = api.index("tb_ctx::WarmTab& wt = ctx->warm_tabs[key];")
    body = api[a:api.index("one table per key, written once", a)]
    import re

    assert "WS_PROLOGUE" in body and "WS_TL_LAYER" in body
    assert not re.search(r"\b\d{4,}\.0\b", body), "a literal cycle count in the warm table"


def test_rollout_source_of_test_step_is_the_whole_history_of_the_batch():
    """Round 6 (tools/fuzz_oracle_vs_reference.py): `test_step` hands the rollout batch["agent/*"] = batch["history/agent/*"] over ALL
    history steps of the batch (`waymo_motion.py:925-926, 538-545`), whatever time_step_current says; `runtime.hist_from_batch` builds
    that source (layout of the agent part of `gt_from_batch`), the teacher-forcing mask over all of its steps included."""
    from trafficbots_amd import synth
    from trafficbots_amd.runtime import hist_from_batch, history_len, scene_from_batch, teacher_forcing_mask

    batch = synth.make_batch(77, 2, n_agent=6, n_pl=8, n_tl=3, p_late_spawn=0.6)
    assert history_len(batch) == 11 and history_len({"agent/valid": np.zeros((1, 91, 2))}) == 0
    hist = hist_from_batch(batch, "cpu", n_hist=6, tf_params=(8, 3))
    scene = scene_from_batch(batch, "cpu", 6)
    assert tuple(scene["agent_valid"].shape) == (2, 6, 6) and tuple(hist["agent_valid"].shape) == (2, 11, 6)
    assert torch.equal(hist["agent_valid"][:, :6], scene["agent_valid"]) and torch.equal(hist["agent_state"][:, :6], scene["agent_state"])
    assert tuple(hist["agent_state"].shape) == (2, 11, 6, 4) and tuple(hist["agent_acc"].shape) == (2, 11, 6)
    want = teacher_forcing_mask(torch.from_numpy(np.asarray(batch["history/agent/valid"])), 8, 3)
    assert torch.equal(hist["_tf_mask"].bool(), want) and hist["_tf_params"] == (8, 3)
    assert want[:, 6:9].any(), "the case should spawn agents beyond step 5: that is what a truncated history would lose"
