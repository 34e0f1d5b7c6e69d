"""ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU restatement (PyTorch-CPU tensor ops, fp32 or fp64) of the reference's scene-encoder +
closed-loop rollout hot path for the default config, written from SURVEY.md Appendix A and the
reference sources cited per function.  It exists to check the HIP path: only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it; the package
`trafficbots_amd` never does (and fails loudly without its HIP library).

Parity pin: validated against the reference itself, imported in the build container, through
the committed fixtures `tests/golden/*.npz` (generator: `tools/gen_golden.py`); see
`tests/test_oracle_golden.py`.

It works on a flat dict of weights in the reference's `state_dict` naming and is purely
functional (no nn.Module), operation order follows the reference so that fp32 results agree
with it to rounding.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F
from torch import Tensor

H = 128
N_HEAD = 4
D_HEAD = 32
LN_EPS = 1e-5


def _t(x, dtype):
    if isinstance(x, np.ndarray):
        x = torch.from_numpy(np.ascontiguousarray(x))
    if x.is_floating_point():
        x = x.to(dtype)
    return x


class Oracle:
    def __init__(self, state_dict: Dict[str, np.ndarray], cfg: dict, dtype=torch.float32, hoist: bool = False,
                 exact_math: bool = False, gemm_order_seed: Optional[int] = None, operand_round: Optional[str] = None,
                 ln_alt: bool = False):
        """`operand_round="bf16"` (VERDICT r03 task 1 (d)): the arithmetic of the library's `operand_precision = "bf16"` mode -- inside the
        ROLLOUT every matrix-product operand of the per-step policy (both operands of the transformer / GRU / fusion / action-head
        Linears that run on the matrix pipe, Q, K, V and the un-normalised softmax weights exp(logit - max)) is rounded to bf16
        (round-to-nearest-even), products and sums stay in `dtype`; LayerNorm, softmax statistics, gate / activation functions, the
        recurrent and simulator state, the small attribute MLP, the 128 -> 2 head and the once-per-rollout `mlp_in` of goal / latent
        stay unrounded, and so do the scene encoders (they keep fp32-accurate operands in either mode).  With dtype=float64 this is the
        "truth" of that arithmetic (rounded operands, exact accumulation).
        `gemm_order_seed` (noise studies only): every Linear sums its products in another order -- the input features and the
        weight columns go through the same random permutation (one per input width) -- so that members of an ensemble sample the
        rounding of the matrix products too, not only that of the attention sums (tools/ensemble.py).
        `exact_math=True` (noise studies only, tests/probes): cos / sin / softmax / sigmoid / tanh of the rollout are evaluated in
        fp64 and rounded once to `dtype` -- another correct implementation of the same fp32 arithmetic, with a different libm.
        `ln_alt=True` (noise studies only, tools/parity_control.py): LayerNorm from explicitly ordered sums -- mean and variance as
        sums over a fixed random channel order, the variance from the centred values -- instead of `F.layer_norm`'s kernel: the same
        function, another association of its two reductions.
        `hoist=False` recomputes the loop-invariant map/TL K,V projections and the goal/latent
        `mlp_in` every step exactly as the reference does; `hoist=True` computes them
        once (same values) -- used where only the result matters."""
        self.dtype = dtype
        self.cfg = cfg
        self.w = {k: _t(v, dtype) for k, v in state_dict.items()}
        self.hoist = hoist
        self.exact_math = exact_math and dtype != torch.float64
        self.ln_alt = ln_alt
        self._ln_perm: Dict[int, Tensor] = {}
        self._kperm: Dict[int, Tensor] = {}
        self._wperm: Dict[tuple, Tensor] = {}
        self._kgen = None if gemm_order_seed is None else torch.Generator().manual_seed(int(gemm_order_seed))
        assert operand_round in (None, "bf16")
        self.operand_round = operand_round
        self._rnd = False  # True while `rollout` runs in operand_round mode
        dyn = cfg["dynamics"]
        # type order veh=0, ped=1, cyc=2 (`dynamics.py:23-27`: instantiate(veh), (ped), (cyc))
        self.max_acc = torch.tensor([dyn["veh"]["max_acc"], dyn["ped"]["max_acc"], dyn["cyc"]["max_acc"]], dtype=dtype)
        self.max_yr = torch.tensor(
            [dyn["veh"]["max_yaw_rate"], dyn["ped"]["max_yaw_rate"], dyn["cyc"]["max_yaw_rate"]], dtype=dtype
        )
        self.dt = 0.1
        self.t_cur = cfg["time_step_current"]
        self.n_hist = self.t_cur + 1

    # ------------------------------------------------------------------ building blocks
    def _m(self, fn, x: Tensor, *a, **k) -> Tensor:
        """transcendental `fn` of x in the working precision (default: what torch's kernels do, as the reference) or via fp64"""
        return fn(x.double(), *a, **k).to(x.dtype) if self.exact_math else fn(x, *a, **k)

    @staticmethod
    def _bf16(x: Tensor) -> Tensor:
        return x.to(torch.bfloat16).to(x.dtype)

    def _linear(self, x: Tensor, w: Tensor, b: Optional[Tensor], mfma: bool = True) -> Tensor:
        """`mfma=False`: a Linear the library evaluates outside the matrix pipe (never operand-rounded)."""
        if self._rnd and mfma:
            x = self._bf16(x)
            key = ("bf16", w.data_ptr(), w.shape[0], w.shape[1])
            if key not in self._wperm:
                self._wperm[key] = self._bf16(w)
            w = self._wperm[key]
        if self._kgen is not None:
            k = w.shape[1]
            if k not in self._kperm:
                self._kperm[k] = torch.randperm(k, generator=self._kgen)
            pm = self._kperm[k]
            key = (w.data_ptr(), w.shape[0], k)  # (weights are never modified: the permuted copy is made once per matrix)
            if key not in self._wperm:
                self._wperm[key] = w[:, pm].contiguous()
            return F.linear(x[..., pm], self._wperm[key], b)
        return F.linear(x, w, b)

    def _ln(self, x: Tensor, prefix: str) -> Tensor:
        if self.ln_alt:
            n = x.shape[-1]
            if n not in self._ln_perm:
                self._ln_perm[n] = torch.randperm(n, generator=torch.Generator().manual_seed(9173 + n))
            xp = x[..., self._ln_perm[n]]
            mean = xp.cumsum(-1)[..., -1:] / n            # (a sequential sum in the permuted order)
            c = x - mean
            var = (c * c)[..., self._ln_perm[n]].cumsum(-1)[..., -1:] / n
            return c * torch.rsqrt(var + LN_EPS) * self.w[prefix + ".weight"] + self.w[prefix + ".bias"]
        return F.layer_norm(x, (x.shape[-1],), self.w[prefix + ".weight"], self.w[prefix + ".bias"], LN_EPS)

    # Linears of the rollout that stay off the matrix pipe in the library (VALU fp32 / once-per-rollout fp32 prologue)
    _NO_MFMA = ("model.agent_encoder.", "model.add_goal.mlp_in.", "model.add_latent.mlp_in.")

    def _lin(self, x: Tensor, prefix: str) -> Tensor:
        off = prefix.startswith(self._NO_MFMA) or (prefix.startswith("action_head.") and prefix.endswith("fc_layers.2"))
        return self._linear(x, self.w[prefix + ".weight"], self.w[prefix + ".bias"], mfma=not off)

    def pose_pe(self, xy: Tensor, yaw: Tensor, who: str = "agent") -> Tensor:
        """`PosePE.forward` mode pe_xy_yaw (`src/utils/pose_pe.py:57-62`) with
        `PositionalEmbedding` / `PositionalEmbeddingRad` (`src/utils/pos_emb.py:24-25,54-55`)."""
        f_xy = self.w[f"pre_processing.input.pose_pe_{who}.pe_xy.freqs"]
        f_yaw = self.w[f"pre_processing.input.pose_pe_{who}.pe_yaw.freqs"]

        def emb(v: Tensor, freqs: Tensor) -> Tensor:
            e = v.unsqueeze(-1) * freqs
            return torch.cat([self._m(torch.cos, e[..., ::2]), self._m(torch.sin, e[..., 1::2])], dim=-1)

        return torch.cat([emb(xy[..., 0], f_xy), emb(xy[..., 1], f_xy), emb(yaw, f_yaw)], dim=-1)

    def input_pe_encoder(self, prefix: str, valid: Tensor, attr: Tensor, pe: Tensor) -> Tensor:
        """`InputPeEncoder.forward`, pe_mode=cat (`input_pe_encoder.py:52-59`)."""
        x = self._lin(torch.relu(self._lin(attr, prefix + ".mlp.fc_layers.0")), prefix + ".mlp.fc_layers.3")
        x = torch.cat([x, pe], dim=-1)
        return x.masked_fill(~valid.unsqueeze(-1), 0)

    def kv_project(self, prefix: str, tgt: Tensor):
        """LN_tgt + packed K,V projection of one layer (`transformer.py:189-192`, `attention.py:81-87`)."""
        w_in = self.w[prefix + ".attn.in_proj_weight"]
        b_in = self.w[prefix + ".attn.in_proj_bias"]
        t = self._ln(tgt, prefix + ".norm_tgt")
        kv = self._linear(t, w_in[H:], b_in[H:])
        return kv[..., :H], kv[..., H:]

    def tf_layer(self, prefix: str, src: Tensor, src_invalid: Tensor, tgt: Optional[Tensor], tgt_invalid: Tensor,
                 attn_mask: Optional[Tensor] = None, kv=None, need_weights: bool = False):
        """(`need_weights`: also the head-mean attention weights [n, ns, nt], zero for rows without an admissible key, `attention.py:142-146`.)
        One pre-LN `TransformerCrossAttention` layer incl. `Attention` (`transformer.py:189-239`,
        `attention.py:81-146`), eval mode.  src [n, ns, H]; tgt [n, nt, H]; masks True = invalid."""
        n, ns, _ = src.shape
        w_in = self.w[prefix + ".attn.in_proj_weight"]
        b_in = self.w[prefix + ".attn.in_proj_bias"]
        s = self._ln(src, prefix + ".norm1")
        q = self._linear(s, w_in[:H], b_in[:H])
        k, v = kv if kv is not None else self.kv_project(prefix, tgt)
        nt = k.shape[-2]
        inv = tgt_invalid.unsqueeze(1).expand(-1, ns, -1)
        if attn_mask is not None:
            inv = inv | attn_mask
        no_tgt = inv.all(-1)  # [n, ns]
        inv = inv & (~no_tgt.unsqueeze(-1))
        q = q.view(n, ns, N_HEAD, D_HEAD).transpose(1, 2)
        k = k.reshape(n, nt, N_HEAD, D_HEAD).transpose(1, 2)
        v = v.reshape(n, nt, N_HEAD, D_HEAD).transpose(1, 2)
        if self._rnd:
            # bf16 operands of the two attention products: Q, K; then the UN-NORMALISED weights e = exp(logit - max) and V, the row sum
            # of the unrounded e dividing afterwards (the library's online softmax normalises once, at the end)
            lg = torch.matmul(self._bf16(q), self._bf16(k).transpose(-2, -1)) / math.sqrt(D_HEAD)
            lg = lg.masked_fill(inv.unsqueeze(1), float("-inf"))
            e = torch.exp(lg - lg.amax(-1, keepdim=True))
            den = e.sum(-1, keepdim=True)
            attn = e / den
            out = (torch.matmul(self._bf16(e), self._bf16(v)) / den).transpose(1, 2).flatten(2, 3)
        else:
            attn = torch.matmul(q, k.transpose(-2, -1))
            attn = attn.masked_fill(inv.unsqueeze(1), float("-inf"))
            attn = self._m(torch.softmax, attn / math.sqrt(D_HEAD), dim=-1)
            out = torch.matmul(attn, v).transpose(1, 2).flatten(2, 3)
        weights = attn.mean(1).masked_fill(no_tgt.unsqueeze(-1), 0) if need_weights else None
        out = self._linear(out, self.w[prefix + ".attn.out_proj_weight"], self.w[prefix + ".attn.out_proj_bias"])
        out = out.masked_fill(no_tgt.unsqueeze(-1), 0)
        src = src + out
        s2 = self._ln(src, prefix + ".norm2")
        s2 = self._lin(torch.relu(self._lin(s2, prefix + ".linear1")), prefix + ".linear2")
        src = src + s2
        src = src.masked_fill(src_invalid.unsqueeze(-1), 0.0)
        return (src, weights) if need_weights else src

    def tf_block(self, prefix: str, n_layer: int, src, src_invalid, tgt, tgt_invalid, attn_mask=None, kvs=None, need_weights: bool = False):
        """`TransformerBlock.forward` (`transformer.py:82-92`): tgt is NOT updated between layers; with `need_weights` the weights of the
        LAST layer come back too (`:81-95`)."""
        w = None
        for i in range(n_layer):
            src = self.tf_layer(f"{prefix}.layers.{i}", src, src_invalid, tgt, tgt_invalid, attn_mask,
                                kv=None if kvs is None else kvs[i], need_weights=need_weights)
            if need_weights:
                src, w = src
        return (src, w) if need_weights else src

    def interaction(self, prefix: str, x: Tensor, valid: Tensor, need_weights: bool = False):
        """`MultiAgentTF.forward` (`agent_interaction.py:51-93`): tgt = block input, eye mask,
        groups with exactly one valid agent pass through unchanged (their weights are zero, `:81-84`)."""
        n, a, _ = x.shape
        eye = torch.eye(a, dtype=torch.bool).unsqueeze(0).expand(n, -1, -1)
        single = valid.sum(-1) == 1
        y = self.tf_block(prefix + ".transformer", 3, x, ~valid, x, ~valid, attn_mask=eye, need_weights=need_weights)
        if need_weights:
            y, w = y
            return torch.where(single[:, None, None], x, y), w.masked_fill(single[:, None, None], 0)
        return torch.where(single[:, None, None], x, y)

    def gru_step(self, prefix: str, x: Tensor, h: Tensor):
        """One time step of `nn.GRU(128,128,3)` (gate order r,z,n).  x [M,H], h [3,M,H]."""
        h_new = []
        inp = x
        for l in range(3):
            gi = self._linear(inp, self.w[f"{prefix}.weight_ih_l{l}"], self.w[f"{prefix}.bias_ih_l{l}"])
            gh = self._linear(h[l], self.w[f"{prefix}.weight_hh_l{l}"], self.w[f"{prefix}.bias_hh_l{l}"])
            i_r, i_z, i_n = gi.chunk(3, -1)
            h_r, h_z, h_n = gh.chunk(3, -1)
            r = self._m(torch.sigmoid, i_r + h_r)
            z = self._m(torch.sigmoid, i_z + h_z)
            nn_ = self._m(torch.tanh, i_n + r * h_n)
            hl = (1 - z) * nn_ + z * h[l]
            h_new.append(hl)
            inp = hl
        return inp, torch.stack(h_new, 0)

    def gru_scan(self, prefix: str, x: Tensor, valid: Tensor) -> Tensor:
        """`MultiAgentGRULoop.forward`, 3-D valid branch (`agent_temporal.py:133-146`).
        x [B,S,A,H], valid [B,S,A] -> [B,S,A,H]."""
        b, s, a, _ = x.shape
        h = torch.zeros(3, b * a, H, dtype=x.dtype)
        outs = []
        for k in range(s):
            o, h = self.gru_step(prefix, x[:, k].reshape(b * a, H), h)
            inv = ~valid[:, k].reshape(b * a, 1)
            h = h.masked_fill(inv.unsqueeze(0), 0.0)
            outs.append(o.masked_fill(inv, 0.0).view(b, a, H))
        return torch.stack(outs, 1)

    # ------------------------------------------------------------------ scene encoders
    def preprocess(self, batch: Dict[str, np.ndarray]) -> Dict[str, Tensor]:
        """`SceneCentricPreProcessing` (eval) + `SceneCentricInput.forward`
        (`scene_centric.py:103-133`, `sc_input.py:100-140`)."""
        d = self.dtype
        nh = self.n_hist
        g = {k: _t(v, d) for k, v in batch.items()}
        out: Dict[str, Tensor] = {}
        av = g["history/agent/valid"][:, :nh]
        out["agent_valid"] = av
        a_type = g["history/agent/type"]
        a_size = g["history/agent/size"]
        n_agent = av.shape[-1]
        out["agent_attr"] = torch.cat(
            [
                g["history/agent/vel"][:, :nh], g["history/agent/spd"][:, :nh], g["history/agent/yaw_rate"][:, :nh],
                g["history/agent/acc"][:, :nh],
                a_size.unsqueeze(1).expand(-1, nh, -1, -1), a_type.unsqueeze(1).expand(-1, nh, -1, -1).to(d),
            ], dim=-1)
        out["agent_pe"] = self.pose_pe(g["history/agent/pos"][:, :nh], g["history/agent/yaw_bbox"][:, :nh, :, 0], "agent")
        out["agent_state"] = torch.cat(
            [g["history/agent/pos"][:, :nh], g["history/agent/yaw_bbox"][:, :nh], g["history/agent/spd"][:, :nh]], -1)
        out["agent_vel"] = g["history/agent/vel"][:, :nh]
        out["agent_acc"] = g["history/agent/acc"][:, :nh]
        out["agent_yaw_rate"] = g["history/agent/yaw_rate"][:, :nh]
        out["agent_type"] = a_type
        out["agent_size"] = a_size
        mv = g["map/valid"]
        b, p, nn_ = mv.shape
        out["map_valid"] = mv
        out["map_attr"] = torch.cat(
            [g["map/type"].unsqueeze(-2).expand(-1, -1, nn_, -1).to(d),
             torch.eye(nn_, dtype=d)[None, None].expand(b, p, -1, -1)], dim=-1)
        map_yaw = torch.atan2(g["map/dir"][..., 1], g["map/dir"][..., 0])
        out["map_pe"] = self.pose_pe(g["map/pos"], map_yaw, "map")
        out["map_type"] = g["map/type"]
        out["map_pos"] = g["map/pos"]
        out["map_dir"] = g["map/dir"]
        out["map_boundary"] = g["map/boundary"]
        out["tl_valid"] = g["history/tl_stop/valid"][:, :nh]
        out["tl_attr"] = g["history/tl_stop/state"][:, :nh].to(d)
        tl_dir = g["history/tl_stop/dir"][:, :nh]
        out["tl_pe"] = self.pose_pe(g["history/tl_stop/pos"][:, :nh], torch.atan2(tl_dir[..., 1], tl_dir[..., 0]), "tl")
        return out

    def preprocess_gt(self, batch: Dict[str, np.ndarray]) -> Dict[str, Tensor]:
        """Validation-split extras: the `latent_post/*` inputs of `SceneCentricLatent.forward` (eval: no perturbation, no
        dropout; `sc_latent.py:150-163,196-217`) and the `gt/*` tensors of `SceneCentricPreProcessing`
        (`scene_centric.py:103-110`), all over the full ground-truth length."""
        d = self.dtype
        g = {k: _t(v, d) for k, v in batch.items() if k.startswith("agent/") or k.startswith("tl_stop/")}
        if "agent/goal" in g:
            out_goal = g["agent/goal"]
        av = g["agent/valid"]
        ns = av.shape[1]
        out: Dict[str, Tensor] = {"agent_valid": av}
        out["agent_attr"] = torch.cat(
            [g["agent/vel"], g["agent/spd"], g["agent/yaw_rate"], g["agent/acc"],
             g["agent/size"].unsqueeze(1).expand(-1, ns, -1, -1), g["agent/type"].unsqueeze(1).expand(-1, ns, -1, -1).to(d)], -1)
        out["agent_pe"] = self.pose_pe(g["agent/pos"], g["agent/yaw_bbox"][..., 0], "agent")
        out["tl_valid"] = g["tl_stop/valid"]
        out["tl_attr"] = g["tl_stop/state"].to(d)
        tl_dir = g["tl_stop/dir"]
        out["tl_pe"] = self.pose_pe(g["tl_stop/pos"], torch.atan2(tl_dir[..., 1], tl_dir[..., 0]), "tl")
        out["agent_state"] = torch.cat([g["agent/pos"], g["agent/yaw_bbox"], g["agent/spd"]], -1)
        out["agent_vel"], out["agent_acc"], out["agent_yaw_rate"] = g["agent/vel"], g["agent/acc"], g["agent/yaw_rate"]
        out["gt_dest"] = g["agent/dest"]
        if "agent/goal" in g:
            out["gt_goal"] = out_goal
        out["agent_role"] = g["agent/role"]
        return out

    def map_encoder(self, inp: Dict[str, Tensor]):
        """`MapEncoder.forward` (`map_encoder.py:72-114`), densetnt_vectornet + max pool."""
        mv = inp["map_valid"]
        b, p, nn_ = mv.shape
        x = self.input_pe_encoder("model.map_encoder.input_pe_encoder", mv, inp["map_attr"], inp["map_pe"])
        x = x.flatten(0, 1)
        v = mv.flatten(0, 1)
        x = self.tf_block("model.map_encoder.transformer_densetnt", 3, x, ~v, x, ~v)
        x = x.view(b, p, nn_, H).masked_fill(~mv.unsqueeze(-1), float("-inf")).amax(dim=2)
        pl_valid = mv.any(-1)
        x = x.masked_fill(~pl_valid.unsqueeze(-1), 0)
        x = self.tf_block("model.map_encoder.transformer_self_attn", 1, x, ~pl_valid, x, ~pl_valid)
        return x, pl_valid

    def encode_scene(self, inp: Dict[str, Tensor]) -> Dict[str, Tensor]:
        """`TrafficBots.encode_input_features` (`traffic_bots.py:147-151`)."""
        f: Dict[str, Tensor] = {}
        f["map_feature"], f["map_valid"] = self.map_encoder(inp)
        f["agent_feature"] = self.input_pe_encoder("model.agent_encoder", inp["agent_valid"], inp["agent_attr"], inp["agent_pe"])
        f["agent_valid"] = inp["agent_valid"]
        f["tl_feature"] = self.input_pe_encoder("model.tl_encoder", inp["tl_valid"], inp["tl_attr"], inp["tl_pe"])
        f["tl_valid"] = inp["tl_valid"]
        return f

    def latent_prior(self, f: Dict[str, Tensor]):
        return self.latent_encode(f, "prior")

    def latent_post(self, f: Dict[str, Tensor], gt: Dict[str, Tensor]):
        """Posterior over the full episode (`waymo_motion.py:583,597`): agent / traffic-light tokens of all ground-truth
        steps, the map feature of the scene."""
        fp = {"map_feature": f["map_feature"], "map_valid": f["map_valid"], "agent_valid": gt["agent_valid"],
              "tl_valid": gt["tl_valid"],
              "agent_feature": self.input_pe_encoder("model.agent_encoder", gt["agent_valid"], gt["agent_attr"], gt["agent_pe"]),
              "tl_feature": self.input_pe_encoder("model.tl_encoder", gt["tl_valid"], gt["tl_attr"], gt["tl_pe"])}
        return self.latent_encode(fp, "post")

    def latent_encode(self, f: Dict[str, Tensor], which: str):
        """`LatentEncoder.forward` (`latent_encoder.py:98-147`; prior and posterior branches differ only in the weights of
        interaction / temporal / head) + `DistEncoder` diag_gaus."""
        r = self.cfg["model"]["latent_encoder"]["temporal_down_sample_rate"]
        av = f["agent_valid"][:, ::r]
        x = f["agent_feature"][:, ::r]
        tv = f["tl_valid"][:, ::r]
        tf_ = f["tl_feature"][:, ::r]
        b, s, a, _ = x.shape
        y = self.tf_block("model.transformer_as2pl", 3, x.flatten(1, 2), ~av.flatten(1, 2), f["map_feature"], ~f["map_valid"])
        y = y.view(b, s, a, H)
        y = self.tf_block("model.transformer_as2tl", 3, y.flatten(0, 1), ~av.flatten(0, 1), tf_.flatten(0, 1), ~tv.flatten(0, 1))
        y = self.interaction(f"model.latent_encoder.agent_interaction_{which}", y, av.flatten(0, 1)).view(b, s, a, H)
        y = self.gru_scan(f"model.latent_encoder.agent_temporal_{which}.rnn", y, av)
        agg = y.masked_fill(~av.unsqueeze(-1), -1e3).amax(1)
        valid = av.any(1)
        agg = agg.masked_fill(~valid.unsqueeze(-1), 0)
        pre = f"model.latent_encoder.latent_{which}_dist"
        mean = self._lin(torch.relu(self._lin(agg, pre + ".mlp_mean.fc_layers.0")), pre + ".mlp_mean.fc_layers.2")
        mean = mean.masked_fill(~valid.unsqueeze(-1), 0)
        return mean, self.w[pre + ".log_std"], valid

    def dest_logits(self, f: Dict[str, Tensor], inp: Dict[str, Tensor]) -> Tensor:
        """`DestPredictor.forward`, mode mlp (`goal_manager.py:229-247,294-307,329-332`).  Returns the
        masked, un-normalised logits [B,A,P]."""
        af, av = f["agent_feature"], f["agent_valid"]
        b, s, a, _ = af.shape
        mtype, atype = inp["map_type"], inp["agent_type"]
        map_type_mask = ~(f["map_valid"] & mtype[:, :, :5].any(-1))
        m_veh = atype[:, :, [0]] & mtype[:, :, 3].unsqueeze(1)
        m_ped = atype[:, :, [1]] & mtype[:, :, :4].any(-1).unsqueeze(1)
        m_cyc = atype[:, :, [2]] & mtype[:, :, :3].any(-1).unsqueeze(1)
        attn_mask = m_veh | m_ped | m_cyc
        dist_valid = av.any(1)
        pre = "model.goal_manager.goal_predictor"
        t = self.gru_scan(pre + ".gru_as.rnn", af, av) + af
        idx_last = s - 1 - torch.max(av.flip(1).to(torch.uint8), dim=1)[1]
        t = t[torch.arange(b).unsqueeze(1), idx_last, torch.arange(a).unsqueeze(0)]
        t = t.masked_fill(~dist_valid.unsqueeze(-1), 0)
        p = f["map_feature"].shape[1]
        pair = torch.cat([f["map_feature"].unsqueeze(1).expand(-1, a, -1, -1), t.unsqueeze(2).expand(-1, -1, p, -1)], -1)
        y = torch.relu(self._ln(self._lin(pair, pre + ".mlp.fc_layers.0"), pre + ".mlp.fc_layers.1"))
        y = torch.relu(self._ln(self._lin(y, pre + ".mlp.fc_layers.3"), pre + ".mlp.fc_layers.4"))
        logits = self._lin(y, pre + ".mlp.fc_layers.6").squeeze(-1)
        logits = logits.masked_fill(map_type_mask.unsqueeze(1), float("-inf"))
        logits = logits.masked_fill(attn_mask, float("-inf"))
        logits = logits.masked_fill(~dist_valid.unsqueeze(-1), 0)
        logits = logits.masked_fill((logits == float("-inf")).all(-1).unsqueeze(-1), 0)
        return logits

    @staticmethod
    def dest_distribution(logits: Tensor, k: int):
        """`DestCategorical` after `repeat_interleave_` (`distributions.py:163-171,197-201`):
        returns (probs [N,A,P], log-prob table [N,A,P]) exactly as torch's Categorical(probs=...) holds them."""
        probs = torch.softmax(logits, -1)  # Categorical(logits).probs
        probs = probs.repeat_interleave(k, 0)
        probs = probs / probs.sum(-1, keepdim=True)
        eps = torch.finfo(probs.dtype).eps
        return probs, torch.log(probs.clamp(min=eps, max=1 - eps))

    # ------------------------------------------------------------------ rollout
    def teacher_forcing_mask(self, valid: Tensor, tf_cfg: dict) -> Tensor:
        """`TeacherForcing.get` (`teacher_forcing.py:45-54`), schedule terms are zero by default."""
        m = torch.zeros_like(valid)
        m[:, 0] |= valid[:, 0]
        ssa = tf_cfg.get("step_spawn_agent", 10)
        if ssa > 0:
            sp = (~valid[:, :-1]) & valid[:, 1:]
            sp[:, ssa:] = False
            m[:, 1:] |= sp
        sws = tf_cfg.get("step_warm_start", 10)
        if sws >= 0:
            m[:, : sws + 1] |= valid[:, : sws + 1]
        return m

    def policy_step(self, x, valid, map_f, map_inv, tl_f, tl_inv, goal_f, goal_valid, z, h, hoisted=None, need_weights: bool = False):
        """`TrafficBots.forward` (`traffic_bots.py:205-241`) for one sim step; with `need_weights` a third return value
        (attn_pl, attn_tl, attn_agent), the reference's `need_weights=True` outputs (`:173,210,218,229`)."""
        w = self.w
        kv_pl = hoisted["kv_pl"] if hoisted else None
        kv_tl = hoisted["kv_tl"] if hoisted else None
        x = self.tf_block("model.transformer_as2pl", 3, x, ~valid, map_f, map_inv, kvs=kv_pl, need_weights=need_weights)
        attn = []
        if need_weights:
            x, w_ = x
            attn.append(w_)
        x = self.tf_block("model.transformer_as2tl", 3, x, ~valid, tl_f, tl_inv, kvs=kv_tl, need_weights=need_weights)
        if need_weights:
            x, w_ = x
            attn.append(w_)
        x = self.interaction("model.agent_interaction", x, valid, need_weights=need_weights)
        if need_weights:
            x, w_ = x
            attn.append(w_)
        n, a, _ = x.shape
        o, h = self.gru_step("model.agent_temporal.rnn", x.reshape(n * a, H), h)
        inv = ~valid.reshape(n * a, 1)
        h = h.masked_fill(inv.unsqueeze(0), 0.0)
        x = o.masked_fill(inv, 0.0).view(n, a, H)
        # add_goal (`add_latent_goal.py:57-77`; MLP masks before the trailing in-place ReLU, `mlp.py:80-84`)
        if hoisted:
            g_pre = hoisted["goal_pre"]
        else:
            g_pre = self.goal_mlp_in(goal_f)
        u = torch.relu(g_pre.masked_fill(~goal_valid.unsqueeze(-1), 0))
        hh = torch.relu(self._lin(torch.cat([x, u], -1), "model.add_goal.mlp_out.fc_layers.0"))
        hh = torch.relu(self._lin(hh, "model.add_goal.mlp_out.fc_layers.3"))
        hh = hh.masked_fill(~goal_valid.unsqueeze(-1), 0) + x
        x = hh.masked_fill(~valid.unsqueeze(-1), 0)
        # add_latent
        z_pre = hoisted["latent_pre"] if hoisted else self.latent_mlp_in(z)
        u = torch.relu(z_pre.masked_fill(~valid.unsqueeze(-1), 0))
        hh = torch.relu(self._lin(torch.cat([x, u], -1), "model.add_latent.mlp_out.fc_layers.0"))
        hh = torch.relu(self._lin(hh, "model.add_latent.mlp_out.fc_layers.3"))
        hh = hh.masked_fill(~valid.unsqueeze(-1), 0) + x
        x = hh.masked_fill(~valid.unsqueeze(-1), 0)
        return (x, h, tuple(attn)) if need_weights else (x, h)

    def goal_mlp_in(self, goal_f: Tensor) -> Tensor:
        p = "model.add_goal.mlp_in.fc_layers"
        y = torch.relu(self._ln(self._lin(goal_f, p + ".0"), p + ".1"))
        y = torch.relu(self._ln(self._lin(y, p + ".4"), p + ".5"))
        return self._ln(self._lin(y, p + ".8"), p + ".9")

    def latent_mlp_in(self, z: Tensor) -> Tensor:
        p = "model.add_latent.mlp_in.fc_layers"
        return self._lin(torch.relu(self._lin(z, p + ".0")), p + ".3")

    def action_mean(self, x: Tensor, valid: Tensor, a_type: Tensor) -> Tensor:
        """`ActionHead.forward`, branch_type (`action_head.py:69-75`)."""
        mean = 0
        for i in range(3):
            y = self._lin(torch.relu(self._lin(x, f"action_head.mlp_mean.{i}.fc_layers.0")), f"action_head.mlp_mean.{i}.fc_layers.2")
            mean = mean + y.masked_fill(~(a_type[:, :, i] & valid).unsqueeze(-1), 0)
        return mean

    def rollout(self, *args, **kwargs) -> Dict[str, Tensor]:
        """`_rollout` with the operand rounding of `operand_round` switched on for its duration (the scene encoders stay unrounded)."""
        self._rnd = self.operand_round == "bf16"
        try:
            return self._rollout(*args, **kwargs)
        finally:
            self._rnd = False

    def _rollout(self, inp: Dict[str, Tensor], f: Dict[str, Tensor], z: Tensor, latent_mean: Tensor, dest: Tensor,
                goal_valid: Tensor, k: int, step_end: int, tap_steps=(), gt: Optional[Dict[str, Tensor]] = None,
                tf_cfg_name: str = "teacher_forcing_joint_future_pred", log_std_name: str = "prior",
                action_eps: Optional[Tensor] = None, action_override=None, hidden_drop=None) -> Dict[str, Tensor]:
        """`hidden_drop` [S] bool: train-mode `p_drop_hidden` (`waymo_motion.py:345-351`) with explicit draws -- after step s (buffer
        written) the GRU state is zeroed where the draw is set.
        `action_override` = (action [N,A,S,2] physical units, mask [N,A,S]): the reference's `forward(action_override=,
        mask_action_override=)` at every step (`dynamics.py:96-100`).
        `action_eps` [N, A, S, 2]: `deterministic_action=False` (`dynamics.py:77`): step s samples the unbounded action as
        mean + eps[:, :, s] * exp(log_std) (`Normal.rsample`) and scores it with `Normal.log_prob` summed over the two dims.
        `WaymoMotion.rollout` + `forward` + `Dynamics` + always-on `TrafficRuleChecker` + buffer
        (`waymo_motion.py:205-354,108-203`, `dynamics.py:50-167,187-228`,
        `traffic_rule_checker.py:77-98,101-119,364-410`), driven as `joint_future_pred` does
        (`waymo_motion.py:516-562`): every scene tensor repeated K times (instance n = b*K + k)."""
        d = self.dtype
        rep = lambda t: t.repeat_interleave(k, 0)  # noqa: E731
        map_f, map_inv = rep(f["map_feature"]), ~rep(f["map_valid"])
        tl_f, tl_v = rep(f["tl_feature"]), rep(f["tl_valid"])
        src = inp if gt is None else gt  # features["agent_valid"/...] of reactive_replay are the 91-step arrays (:457-461)
        hv = rep(src["agent_valid"])
        hs, hvel, hacc, hyr = rep(src["agent_state"]), rep(src["agent_vel"]), rep(src["agent_acc"]), rep(src["agent_yaw_rate"])
        a_type, a_size = rep(inp["agent_type"]), rep(inp["agent_size"])
        n, a = dest.shape
        mask_tf = self.teacher_forcing_mask(hv, self.cfg[tf_cfg_name])
        n_gt = hv.shape[1]
        # rule-checker geometry gathered once (`traffic_rule_checker.py:85-98`)
        bidx = torch.arange(n).unsqueeze(1)
        mvalid, mtype, mpos, mdir = rep(inp["map_valid"]), rep(inp["map_type"]), rep(inp["map_pos"]), rep(inp["map_dir"])
        bound = rep(inp["map_boundary"])
        dest_valid = mvalid[bidx, dest]
        dest_type = mtype[bidx, dest]
        dest_pos = mpos[bidx, dest]
        dest_dir = mdir[bidx, dest]
        dest_dir = dest_dir / torch.norm(dest_dir, dim=-1, keepdim=True)
        thresh_rot_cos = np.cos(np.deg2rad(30))
        thresh_pos = torch.ones_like(a_size[:, :, 0]) * 50
        thresh_pos = thresh_pos * (1 - dest_type[:, :, 4] * 0.8)
        mask_lane = dest_type[:, :, :4].any(-1)
        mask_edge = dest_type[:, :, 4]
        # latent log-prob (`distributions.py:11-15`, Independent(Normal))
        log_std = self.w[f"model.latent_encoder.latent_{log_std_name}_dist.log_std"]
        std = log_std.exp()
        lm = rep(latent_mean)
        latent_logp = (-((z - lm) ** 2) / (2 * std ** 2) - std.log() - math.log(math.sqrt(2 * math.pi))).sum(-1)
        goal_f = map_f[bidx, dest]
        hoisted = None
        if self.hoist:
            hoisted = {
                "kv_pl": [self.kv_project(f"model.transformer_as2pl.layers.{i}", map_f) for i in range(3)],
                "goal_pre": self.goal_mlp_in(goal_f),
                "latent_pre": self.latent_mlp_in(z),
            }
            kv_tl_all = [[self.kv_project(f"model.transformer_as2tl.layers.{i}", tl_f[:, s]) for i in range(3)]
                         for s in range(tl_f.shape[1])]
        # dynamics init (`dynamics.py:29-48`)
        valid = hv[:, 0].clone()
        killed = torch.zeros_like(valid)
        state = hs[:, 0].clone()
        vel, acc, yr = hvel[:, 0].clone(), hacc[:, 0].clone(), hyr[:, 0].clone()
        h = torch.zeros(3, n * a, H, dtype=d)
        outside = torch.zeros_like(valid)
        dest_reached = torch.zeros_like(valid)
        goal_pose = rep(gt["gt_goal"]) if (gt is not None and "gt_goal" in gt) else None  # traffic_rule_checker.py:473-479
        goal_reached = torch.zeros_like(valid)
        atype_idx = a_type.to(torch.int64).argmax(-1)
        max_acc = self.max_acc[atype_idx]
        max_yr = self.max_yr[atype_idx]
        has_type = a_type.any(-1)
        out = {k_: [] for k_ in ("preds", "valid", "override_masks", "outside_map", "outside_map_this_step",
                                 "dest_reached", "dest_reached_this_step", "action_log_probs", "goal_reached",
                                 "goal_reached_this_step", "check_state", "check_valid")}
        taps: Dict[str, Tensor] = {}
        for t in range(self.cfg["time_step_sim_start"], step_end + 1):
            m_ovr = mask_tf[:, t] if t < n_gt else torch.zeros_like(valid)
            step_tl = min(t - 1, tl_v.shape[1] - 1)
            # ---- WaymoMotion.forward ----
            attr = torch.cat([vel, state[:, :, 3:4], yr, acc, a_size, a_type.to(d)], -1)
            pe = self.pose_pe(state[:, :, :2], state[:, :, 2], "agent")
            x = self.input_pe_encoder("model.agent_encoder", valid, attr, pe)
            if t in tap_steps:
                taps[f"tap{t}/agent_feature"] = x.clone()
                taps[f"tap{t}/state_in"] = state.clone()
                taps[f"tap{t}/agent_valid"] = valid.clone()
                taps[f"tap{t}/goal_valid"] = goal_valid.clone()
            if hoisted is not None:
                hoisted["kv_tl"] = kv_tl_all[step_tl]
            x, h = self.policy_step(x, valid, map_f, map_inv, tl_f[:, step_tl], ~tl_v[:, step_tl], goal_f, goal_valid, z, h, hoisted)
            if t in tap_steps:
                taps[f"tap{t}/policy_feature"] = x.clone()
                taps[f"tap{t}/hidden"] = h.clone()
            u = self.action_mean(x, valid, a_type)
            # action_head.py:81-87: per-type log_std (a PARAMETER of the state_dict) under (type & valid), 0 elsewhere -- an agent
            # without a type is scored with log_std = 0 (golden `edge_scenes`); distributions.py:49
            tmask = a_type & valid.unsqueeze(-1)
            ls = sum(self.w[f"action_head.log_std.{i}"].to(d)[None, None, :] * tmask[:, :, [i]].to(d) for i in range(3))
            dist = torch.distributions.Independent(torch.distributions.Normal(u, ls.exp()), 1)
            if action_eps is None:
                alp = dist.log_prob(u).masked_fill(~valid, 0)  # (Normal(mu, sigma).log_prob(mu): no dependence on mu)
            else:
                u = u + _t(action_eps, d)[:, :, t - self.cfg["time_step_sim_start"]] * ls.exp()
                alp = dist.log_prob(u).masked_fill(~valid, 0)
            act = self._m(torch.tanh, u)
            act = torch.stack([act[..., 0] * max_acc, act[..., 1] * max_yr], -1)
            act = act.masked_fill(~(has_type & valid).unsqueeze(-1), 0)
            if action_override is not None:  # dynamics.py:96-100 (after the per-type masking; only for valid agents)
                s_i = t - self.cfg["time_step_sim_start"]
                am = torch.as_tensor(action_override[1][:, :, s_i]).bool() & valid
                act = torch.where(am.unsqueeze(-1), _t(action_override[0][:, :, s_i], d), act)
            # MultiPathPP.update (`dynamics.py:208-221`)
            a_, w_ = act[:, :, 0], act[:, :, 1]
            v_t = state[:, :, 3] + 0.5 * self.dt * a_
            th_t = state[:, :, 2] + 0.5 * self.dt * w_
            delta = torch.stack([v_t * self._m(torch.cos, th_t), v_t * self._m(torch.sin, th_t), w_, a_], -1)
            new_state = (state + self.dt * delta).masked_fill(~(has_type & valid).unsqueeze(-1), 0)
            pred_state, pred_valid = new_state, valid
            state = new_state
            # override_states (`dynamics.py:132-149`)
            m = m_ovr & (~killed)
            if m.any():
                valid = valid | m
                m3 = m.unsqueeze(-1)
                state = torch.where(m3, hs[:, t], state)
                vel = torch.where(m3, hvel[:, t], vel)
                acc = torch.where(m3, hacc[:, t], acc)
                yr = torch.where(m3, hyr[:, t], yr)
            # rule check (`traffic_rule_checker.py:101-119,364-410`)
            check_state, check_valid = state.clone(), valid.clone()  # what TrafficRuleChecker.check is handed (:311)
            xs, ys = state[:, :, 0], state[:, :, 1]
            out_this = ((xs > bound[:, [1]]) | (xs < bound[:, [0]]) | (ys > bound[:, [3]]) | (ys < bound[:, [2]])) & valid
            outside = outside | out_this
            dist = torch.norm(state[..., :2].unsqueeze(2) - dest_pos, dim=-1).masked_fill(~dest_valid, 1e4)
            pos_reached = (dist < thresh_pos.unsqueeze(-1)).any(-1)
            head = torch.stack([torch.cos(state[..., 2]), torch.sin(state[..., 2])], -1)
            rot = (head.unsqueeze(2) * dest_dir).sum(-1).masked_fill(~dest_valid, 0)
            rot_reached = (rot > thresh_rot_cos).any(-1)
            dr_this = (~dest_reached) & valid & ((mask_lane & pos_reached & rot_reached) | (mask_edge & pos_reached))
            dest_reached = dest_reached | dr_this
            gr_this = torch.zeros_like(valid)
            if goal_pose is not None:  # _check_goal_reached (:337-361): 8 agent lengths, 15 degrees
                pos_ok = torch.norm(state[..., :2] - goal_pose[..., :2], dim=-1) < a_size[:, :, 0] * 8
                dyaw = (state[..., 2] - goal_pose[..., 2] + math.pi) % (2 * math.pi) - math.pi
                gr_this = pos_ok & (dyaw.abs() < float(np.deg2rad(15))) & valid & (~goal_reached)
            goal_reached = goal_reached | gr_this
            # kill (`dynamics.py:161-167`)
            mk = out_this & (~hv[:, t]) if t < n_gt else out_this
            killed = killed | mk
            valid = valid & (~mk)
            # navigator (`goal_manager.py:155-162`)
            goal_valid = goal_valid & valid & (~dest_reached)
            out["preds"].append(pred_state)
            out["valid"].append(pred_valid)
            out["override_masks"].append(m_ovr)
            out["outside_map"].append(outside)
            out["outside_map_this_step"].append(out_this)
            out["dest_reached"].append(dest_reached)
            out["dest_reached_this_step"].append(dr_this)
            out["action_log_probs"].append(alp)
            out["goal_reached"].append(goal_reached)
            out["check_state"].append(check_state)
            out["check_valid"].append(check_valid)
            out["goal_reached_this_step"].append(gr_this)
            if hidden_drop is not None and bool(hidden_drop[t - self.cfg["time_step_sim_start"]]):
                h = torch.zeros_like(h)
        res = {k_: torch.stack(v, 2) for k_, v in out.items()}
        s = res["valid"].shape[2]
        res["latent_log_probs"] = latent_logp.unsqueeze(-1).expand(-1, -1, s)
        res["final_state"], res["final_valid"], res["final_hidden"] = state, valid, h
        res["final_goal_valid"] = goal_valid
        res.update(taps)
        return res

    # ------------------------------------------------------------------ end to end
    def joint_future_pred(self, batch: Dict[str, np.ndarray], k: int, eps: Optional[np.ndarray], step_end: int,
                          dest_override: Optional[np.ndarray] = None, tap_steps=(), use_gt: bool = False,
                          action_eps: Optional[np.ndarray] = None, action_override=None) -> Dict[str, Tensor]:
        """Prologue of `WaymoMotion.test_step` + `joint_future_pred` (`waymo_motion.py:902-933,478-572`).
        `eps` [N,A,16] are the standard-normal draws of the latent sample; `dest_override` [N,A]
        replaces the destination sample (instances k>0 are multinomial draws in the reference)."""
        inp = self.preprocess(batch)
        f = self.encode_scene(inp)
        mean, log_std, lvalid = self.latent_prior(f)
        logits = self.dest_logits(f, inp)
        probs, logp_table = self.dest_distribution(logits, k)
        b, a = mean.shape[:2]
        n = b * k
        det = torch.zeros(n, a, dtype=torch.bool)
        det[::k] = True
        mean_r = mean.repeat_interleave(k, 0)
        if eps is None:
            z = mean_r.clone()
        else:
            rnd = mean_r + _t(eps, self.dtype) * log_std.exp()
            z = mean_r.masked_fill(~det.unsqueeze(-1), 0) + rnd.masked_fill(det.unsqueeze(-1), 0)
        dest = probs.argmax(-1)
        if dest_override is not None:
            dest = _t(dest_override, self.dtype).to(torch.int64).reshape(n, a)
        goal_logp = logp_table.gather(-1, dest.unsqueeze(-1)).squeeze(-1)
        goal_valid = inp["agent_valid"].any(1).repeat_interleave(k, 0)
        # validation_step leaves batch["agent/*"] = the 91-step ground truth in place (waymo_motion.py:538-545), which the kill
        # rule and goal_reached then see; test_step overwrites it with the history (:925-926)
        gt = self.preprocess_gt(batch) if use_gt else None
        if gt is None and np.asarray(batch["history/agent/valid"]).shape[1] > self.n_hist:
            # test_step hands the rollout batch["agent/*"] = batch["history/agent/*"] over ALL history steps of the batch
            # (waymo_motion.py:925-926, 538-545) while the encoders see the first time_step_current + 1 of them (scene_centric.py:92-121):
            # with time_step_current < 10 the teacher-forcing mask, the state overrides and the kill rule's ground-truth validity still
            # reach to the end of the 11-step history (tools/fuzz_oracle_vs_reference.py cases 74, 118)
            d_ = self.dtype
            hg = lambda k_: _t(batch["history/agent/" + k_], d_)  # noqa: E731
            gt = {"agent_valid": _t(batch["history/agent/valid"], d_), "agent_vel": hg("vel"), "agent_acc": hg("acc"),
                  "agent_yaw_rate": hg("yaw_rate"), "agent_state": torch.cat([hg("pos"), hg("yaw_bbox"), hg("spd")], -1)}
        res = self.rollout(inp, f, z, mean, dest, goal_valid, k, step_end, tap_steps, gt=gt, action_eps=action_eps,
                           action_override=action_override)
        # RolloutBuffer.flatten_repeat (`buffer.py:92-123`): [N,A,S,..] -> [B,A,K,S,..]
        for key in ("preds", "valid", "override_masks", "outside_map", "outside_map_this_step", "dest_reached",
                    "dest_reached_this_step", "action_log_probs", "latent_log_probs", "goal_reached", "goal_reached_this_step"):
            v = res[key]
            res[key] = v.reshape(b, k, *v.shape[1:]).transpose(1, 2)
        res["goal_sample"] = dest.view(b, k, a).transpose(1, 2)
        res["goal_log_probs"] = goal_logp.view(b, k, a).transpose(1, 2)
        res["dest_logits_raw"] = logits
        res["dest_logits"] = torch.log_softmax(logits, -1)
        res["latent_mean"], res["latent_valid"], res["latent_sample"] = mean, lvalid, z
        res["map_feature"], res["map_feature_valid"] = f["map_feature"], f["map_valid"]
        res["agent_feature_cur"], res["agent_feature_0"] = f["agent_feature"][:, -1], f["agent_feature"][:, 0]
        res["tl_feature_cur"] = f["tl_feature"][:, -1]
        res["_inp"], res["_feats"] = inp, f
        return res

    def reactive_replay(self, batch: Dict[str, np.ndarray], step_end: int, tf_cfg_name: str = "teacher_forcing_reactive_replay",
                        eps: Optional[np.ndarray] = None, rollout_prior: bool = False,
                        action_eps: Optional[np.ndarray] = None, history_keep: Optional[Dict[str, np.ndarray]] = None,
                        hidden_drop: Optional[np.ndarray] = None) -> Dict[str, Tensor]:
        """`history_keep` / `hidden_drop`: the train-mode Bernoulli masks with explicit draws (synth.make_train_draws): KEEP masks of the
        model inputs (`sc_input.py:100-106`: agent history but its last step, traffic lights, map nodes) and of the posterior's inputs
        (`sc_latent.py:171-173,216-218`), hidden-state drop after a step (`waymo_motion.py:345-351`).  The posterior's masks are applied
        IN PLACE to `batch["agent/valid"]` / `batch["tl_stop/valid"]` in the reference (`latent_post/agent_valid = batch["agent/valid"]`
        is an alias, `sc_latent.py:215-218`), so the replay's ground-truth validity -- initial state, teacher-forcing mask, kill rule,
        rewards -- is the MASKED one too; reproduced here.
        First half of `WaymoMotion.validation_step` (`waymo_motion.py:574-611`): posterior and prior personalities,
        destination prediction, then the episode replayed with the posterior MEAN as personality, the ground-truth
        destination as goal and `teacher_forcing_reactive_replay` (agents keep spawning from ground truth until step 90).
        Outputs are the un-flattened RolloutBuffer fields [B,A,S,...] plus what `TrainingMetrics.update` consumes."""
        inp = self.preprocess(batch)
        gt = self.preprocess_gt(batch)
        inp_enc, gt_enc = inp, gt
        if history_keep is not None:
            kb = lambda k_: torch.as_tensor(np.asarray(history_keep[k_])).bool()  # noqa: E731
            av = inp["agent_valid"].clone()
            av[:, :-1] &= kb("input_agent")
            inp_enc = dict(inp, agent_valid=av, tl_valid=inp["tl_valid"] & kb("input_tl"), map_valid=inp["map_valid"] & kb("input_map"))
            gt_enc = dict(gt, agent_valid=gt["agent_valid"] & kb("post_agent"), tl_valid=gt["tl_valid"] & kb("post_tl"))
            if av.shape[0] == 1:
                # A batch of ONE scene (found by tools/fuzz_oracle_vs_reference.py --train with n_scene = 1): `sc/agent_valid =
                # batch["agent/valid"][:, :n_step_hist].contiguous()` (scene_centric.py:92-99; likewise tl) is a VIEW when the batch
                # dimension is 1 -- the slice is already contiguous -- so the history tensors and the first steps of the ground truth share
                # storage and every in-place `&=` (sc_input.py:105-113, sc_latent.py:171-173,216-218) lands on both: the history carries
                # the posterior's masks, the ground truth's first steps the input's.  With two or more scenes the slice is a copy.
                nh = av.shape[1]
                av_all = inp_enc["agent_valid"] & gt_enc["agent_valid"][:, :nh]
                tl_all = inp_enc["tl_valid"] & gt_enc["tl_valid"][:, :nh]
                inp_enc = dict(inp_enc, agent_valid=av_all, tl_valid=tl_all)
                ga, gtl = gt_enc["agent_valid"].clone(), gt_enc["tl_valid"].clone()
                ga[:, :nh], gtl[:, :nh] = av_all, tl_all
                gt_enc = dict(gt_enc, agent_valid=ga, tl_valid=gtl)
        f = self.encode_scene(inp_enc)
        post_mean, post_log_std, post_valid = self.latent_post(f, gt_enc)
        prior_mean, prior_log_std, prior_valid = self.latent_prior(f)
        logits = self.dest_logits(f, inp_enc)
        goal_valid = inp_enc["agent_valid"].any(1)  # GoalManager.get_gt_goal, goal_manager.py:66-68
        dest = gt["gt_dest"].to(torch.int64)
        # training_step (`waymo_motion.py:384-400`): a SAMPLE (mean + std * eps) of the posterior, or of the prior
        mean, log_std, which = (prior_mean, prior_log_std, "prior") if rollout_prior else (post_mean, post_log_std, "post")
        z = mean.clone() if eps is None else mean + _t(eps, self.dtype) * log_std.exp()
        # (`input/map_valid` IS `map/valid` in the reference -- `sc/map_valid = batch["map/valid"]`, scene_centric.py:113-114, masked in
        # place by sc_input.py:105-106 -- so the rule checker's destination geometry sees the masked nodes too)
        inp_roll = inp if history_keep is None else dict(inp, map_valid=inp_enc["map_valid"])
        res = self.rollout(inp_roll, f, z, mean, dest, goal_valid, 1, step_end, gt=gt_enc, tf_cfg_name=tf_cfg_name, log_std_name=which,
                           action_eps=action_eps, hidden_drop=hidden_drop)
        res.update(post_mean=post_mean, post_log_std=post_log_std, post_valid=post_valid, prior_mean=prior_mean,
                   prior_log_std=prior_log_std, prior_valid=prior_valid, dest_logits_raw=logits, goal_valid=goal_valid,
                   gt_dest=dest, gt_valid=gt_enc["agent_valid"], gt_state=gt["agent_state"], agent_role=gt["agent_role"],
                   agent_size=inp["agent_size"], agent_type=inp["agent_type"])
        res["_inp"], res["_feats"], res["_gt"] = inp, f, gt
        return res
