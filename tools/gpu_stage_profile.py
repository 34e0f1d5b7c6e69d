"""Development probe: per-stage cycle breakdown of one fused k_step launch from the -DTB_PROFILE build
(s_memtime stamps written by thread 0 of every workgroup).  Run on the GPU box:
    TB_HIP_LIB=trafficbots_amd/lib/libtrafficbots_hip_prof.so python tools/gpu_stage_profile.py"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from trafficbots_amd import synth  # noqa: E402
from trafficbots_amd.config import load_model_config  # noqa: E402
from trafficbots_amd.runtime import HipEngine, scene_from_batch  # noqa: E402

NAMES = ["C: tile/geometry loads", "C: interaction x3", "C: GRU x3", "C: add_goal", "C: add_latent", "C: action head",
         "C: epilogue", "A: attr+PE+encoder", "A: as2pl x3", "A: as2tl x3", "A: interaction K/V proj x3"]

cfg = load_model_config(overrides={"operand_precision": os.environ.get("TB_PRECISION", "fp32")})
sd = synth.make_state_dict(7)
b, a, p, t = 32, int(os.environ.get("AB_A", 64)), int(os.environ.get("AB_P", 256)), 40
batch = synth.make_batch(5000, b, n_agent=a, n_pl=p, n_tl=t)
dev = torch.device("cuda:0")
eng = HipEngine(cfg)
eng.load_state_dict(sd)
s = scene_from_batch(batch, dev)
enc = eng.encode_scene(s)
feats = {"map_feature": enc["map_feature"], "map_feature_valid": enc["map_feature_valid"], "tl_feature": enc["tl_feature"]}
k = int(os.environ.get("AB_K", 1))  # futures per scene (BASELINE configs[3]: 6)
z = enc["latent_mean"].repeat_interleave(k, 0).contiguous()
if k > 1:
    z = z + 0.3 * torch.from_numpy(synth.make_latent_noise(1, b * k, a)).to(z.device)
dest = enc["dest_logits"].argmax(-1).to(torch.int32).repeat_interleave(k, 0).contiguous()
gv = s["agent_valid"].bool().any(1).to(torch.uint8).repeat_interleave(k, 0).contiguous()
out = None
for step_end in (90, 50):  # the stamps kept are those of the LAST launch that ran both halves: use step_end-1 ... see below
    out = eng.rollout(s, feats, z, enc["latent_mean"], dest, gv, k, step_end)
torch.cuda.synchronize()
# last launch = C(S) only; stamps 0..7 of it are overwritten, stamps 8..11 are those of the previous (fused) launch
n_blocks = b * k * (a // 16)
buf = (C.c_longlong * (32 * n_blocks))()
eng.lib.tb_debug_read_prof.argtypes = [C.c_void_p, C.POINTER(C.c_longlong), C.c_int]
rc = eng.lib.tb_debug_read_prof(eng._ctx, buf, n_blocks)
assert rc == 0
st = np.frombuffer(buf, dtype=np.int64).reshape(n_blocks, 32)
d_c = np.diff(st[:, 0:8], axis=1)     # C half of the last launch
d_a = np.diff(st[:, 7:12], axis=1)    # NOTE: stamp 7 is from the last launch, 8.. from the previous one -> use 8..11 only
d_a = np.diff(st[:, 8:12], axis=1)
print(f"workgroups: {n_blocks}; cycles are s_memtime ticks (median over workgroups [min..max])")
tot = 0
for i in range(7):
    v = d_c[:, i]
    print(f"  {NAMES[i]:32s} {np.median(v):9.0f}  [{v.min():8.0f} .. {v.max():8.0f}]")
    tot += np.median(v)
for i in range(3):
    v = d_a[:, i]
    print(f"  {NAMES[8 + i]:32s} {np.median(v):9.0f}  [{v.min():8.0f} .. {v.max():8.0f}]")
    tot += np.median(v)
v = st[:, 8] - st[:, 30]
print(f"  {NAMES[7]:32s} {np.median(v):9.0f}  [{v.min():8.0f} .. {v.max():8.0f}]   (k_step_x only)")
print(f"  sum of medians (without A front-end) {tot:9.0f}")
if os.environ.get("TB_STEP_KERNEL") != "fp32":  # k_step_x: stamps inside the one-burst prologue (round 5)
    seqx = [31, 0, 12, 13, 14, 15, 1]
    labx = ["kernel entry -> tile map, first kernarg use", "first weight unit issued (addresses from kernarg)", "all prologue loads issued",
            "encoder weights + epilogue record in LDS (first waits)", "C tiles + LN blocks in LDS (all loads back)", "barrier"]
    for i in range(6):
        v = st[:, seqx[i + 1]] - st[:, seqx[i]]
        print(f"    C-start/{labx[i]:52s} {np.median(v):8.0f}  [{v.min():7.0f} .. {v.max():7.0f}]")
    if a * b * k // 16 <= 128:  # helper workgroups (slots 28 / 29: their entry / end of their own work; the last launch is C only: same grid)
        e_t, e_h, d_h, c_end = st[:, 27], st[:, 28], st[:, 29], st[:, 26]  # s_memrealtime ticks (10 ns)
        t0 = min(e_t.min(), e_h.min())
        us = lambda v: 0.01 * float(v)
        print(f"    dispatch (us since the first workgroup's entry): helper entries {us(e_h.min() - t0):.2f} .. {us(e_h.max() - t0):.2f}, tile entries "
              f"{us(e_t.min() - t0):.2f} .. {us(e_t.max() - t0):.2f} (median {us(np.median(e_t) - t0):.2f}); helpers done {us(np.median(d_h) - t0):.2f} (median) "
              f"{us(d_h.max() - t0):.2f} (last); tiles' C half done {us(np.median(c_end) - t0):.2f} (median) {us(c_end.max() - t0):.2f} (last)")
seq = [0, 12, 13, 14, 15, 1]
lab = ["LN params -> LDS (issue+store)", "row state loads (16 thr)", "valid ballots", "wload + 6 tile loads + geometry issue", "barrier (wait for all)"]
if os.environ.get("TB_STEP_KERNEL") == "fp32":  # (k_step_x keeps no stamps inside its one-round-trip prologue)
    for i in range(5):
        v = st[:, seq[i + 1]] - st[:, seq[i]]
        print(f"    C-start/{lab[i]:40s} {np.median(v):8.0f}  [{v.min():7.0f} .. {v.max():7.0f}]")
lab = ["LN1 + barrier", "wload + Q proj (64 MFMA)", "attention 256 keys (8 x 32 MFMA)", "store + barrier", "wload + out proj + residual + barrier",
       "LN2 + barrier", "wload + FFN1 + relu + barrier", "wload + FFN2 + residual + barrier"]
for i in range(8):
    v = st[:, 17 + i] - st[:, 16 + i]
    print(f"    as2pl[0]/{lab[i]:40s} {np.median(v):8.0f}  [{v.min():7.0f} .. {v.max():7.0f}]")
for nm, a_, b_ in (("blk0 phase A (QK next || exp)", 25, 26), ("blk0 phase B (PV || stats next)", 26, 27), ("blk1 phase A", 27, 28), ("blk1 phase B", 28, 29)):
    v = st[:, b_] - st[:, a_]
    print(f"    as2pl[0]/attention {nm:34s} {np.median(v):8.0f}  [{v.min():7.0f} .. {v.max():7.0f}]")
print(f"    as2pl[0]/attention prologue (loads + first QK + stats) {np.median(st[:, 25] - st[:, 18]):8.0f}")
if os.environ.get("TB_STEP_AW", "1") != "0" and a * b // 16 > 128 and p >= 512 and os.environ.get("TB_PRECISION") == "bf16":
    # the assist carve (tb::xba): stamp 25 = the main wave's own half of the walk is done (in front of the merge barrier)
    v, w = st[:, 25] - st[:, 18], st[:, 19] - st[:, 25]
    print(f"    as2pl[0]/assist carve: main wave 0's own blocks {np.median(v):8.0f} [{v.min():7.0f} .. {v.max():7.0f}]; merge barrier + merge {np.median(w):8.0f} [{w.min():7.0f} .. {w.max():7.0f}]")

# ---- machine-readable constants for bench.py::structural_floor (the floor used round-3 numbers)
if os.environ.get("TB_STAGE_JSON"):
    import json

    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge

    med = lambda v: float(np.median(v))  # noqa: E731
    rec = {
        "what": "s_memtime cycles of one fused k_step_x launch, -DTB_PROFILE build of the sources with this fingerprint (medians over workgroups)",
        "src_sha256": ge.build_fingerprint(), "precision": os.environ.get("TB_PRECISION", "fp32"), "shape": {"B": b, "K": k, "A": a, "P": p, "T": t},
        "workgroups": int(n_blocks),
        "c_half": {NAMES[i]: med(d_c[:, i]) for i in range(7)},
        "a_half": {NAMES[8 + i]: med(d_a[:, i]) for i in range(3)},
        "a_front_end": med(st[:, 8] - st[:, 30]),
        "cold_start_cycles": med(d_c[:, 0]),
        "as2pl0": {lab[i]: med(st[:, 17 + i] - st[:, 16 + i]) for i in range(8)},
        "sum_of_medians": float(tot),
    }
    # the serial chains the weight stream cannot hide: the attention walks (as2pl layer 0's walk x 3 layers, scaled by keys for the
    # traffic-light and interaction walks) and the 21 LayerNorms (as2pl layer 0's two)
    walk = rec["as2pl0"]["attention 256 keys (8 x 32 MFMA)"]
    rec["serial_chains_cycles"] = {"attention_walks": 3 * walk + 3 * walk * (32.0 / p) * 2 + 3 * walk * (a / p),
                                   "layernorms_21": 21 * 0.5 * (rec["as2pl0"]["LN1 + barrier"] + rec["as2pl0"]["LN2 + barrier"]),
                                   "how": "3 map walks measured on as2pl[0]; traffic-light / interaction walks scaled by their key counts; 21 LayerNorms at the mean of as2pl[0]'s two"}
    with open(os.environ["TB_STAGE_JSON"], "w") as f:
        json.dump(rec, f, indent=1)
    print("wrote", os.environ["TB_STAGE_JSON"])
