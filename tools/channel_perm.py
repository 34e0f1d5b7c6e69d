"""Mathematically equivalent re-labellings of the model's hidden channels, for ensemble members whose GEMM sums are re-ordered too.

A batch permutation (tools/ensemble.py) re-orders the attention / pooling sums of the reference but leaves every `F.linear` summing its
128 products in the base run's order, so such members share the base run's GEMM rounding.  Here the 128 channels of
every hidden tensor of the network are re-labelled: columns of every consumer Linear, rows + bias of every producer, LayerNorm gamma / beta,
gate blocks of the GRUs, whole heads and channels within a head for the attention projections.  The permuted `state_dict` loaded into
the imported reference, plus a forward hook that re-labels the output of the three `InputPeEncoder`s (their 96 positional-encoding
channels are computed, not learned), is the same function of the inputs -- checked in fp64 by tools/gen_golden_ensg.py (<= 1e-9 m over the
whole rollout) -- with another summation order inside every Linear, LayerNorm and attention product.

Convention: a layout `p` (int array [128]) means new_vec[j] = old_vec[p[j]].  A Linear y = W x + b with input layout p_in and output layout
p_out becomes W[p_out][:, p_in], b[p_out].

Test infrastructure only (tools/gen_golden*.py with the imported reference).  Module structure: SURVEY.md Appendix A / B;
reference `src/models/traffic_bots.py:163-247`, `src/models/latent_encoder.py:70-147`, `src/models/goal_manager.py:202-333`,
`src/models/modules/{transformer,attention,add_latent_goal,action_head,agent_temporal,map_encoder}.py`.
"""
from __future__ import annotations

from typing import Dict, Set

import numpy as np

H, N_HEAD = 128, 4
D_HEAD = H // N_HEAD


class _Ctx:
    def __init__(self, sd: Dict[str, np.ndarray], seed: int):
        self.src = sd
        self.out: Dict[str, np.ndarray] = {}
        self.done: Set[str] = set()
        self.rng = np.random.Generator(np.random.PCG64(seed))
        self.ident = np.arange(H)

    def free(self, n: int = H) -> np.ndarray:
        return self.rng.permutation(n)

    def heads(self, sigma=None) -> np.ndarray:
        """Whole heads permuted by `sigma` (drawn if None), channels permuted within each head."""
        sigma = self.rng.permutation(N_HEAD) if sigma is None else sigma
        return np.concatenate([sigma[h] * D_HEAD + self.rng.permutation(D_HEAD) for h in range(N_HEAD)]), sigma

    def put(self, key: str, value: np.ndarray) -> None:
        assert key in self.src and key not in self.done, key
        assert value.shape == self.src[key].shape, key
        self.out[key] = np.ascontiguousarray(value)
        self.done.add(key)

    def linear(self, pfx: str, p_out, p_in) -> None:
        w, b = self.src[pfx + ".weight"], self.src[pfx + ".bias"]
        w = w if p_out is None else w[p_out]
        self.put(pfx + ".weight", w if p_in is None else w[:, p_in])
        self.put(pfx + ".bias", b if p_out is None else b[p_out])

    def norm(self, pfx: str, p) -> None:
        self.put(pfx + ".weight", self.src[pfx + ".weight"][p])
        self.put(pfx + ".bias", self.src[pfx + ".bias"][p])

    def transformer(self, pfx: str, n_layer: int, p_src, p_tgt) -> None:
        for i in range(n_layer):
            l = f"{pfx}.layers.{i}"
            self.norm(l + ".norm1", p_src)
            self.norm(l + ".norm_tgt", p_tgt)
            self.norm(l + ".norm2", p_src)
            p_qk, sigma = self.heads()
            p_v, _ = self.heads(sigma)
            w, b = self.src[l + ".attn.in_proj_weight"], self.src[l + ".attn.in_proj_bias"]
            wq, wk, wv = w[:H], w[H : 2 * H], w[2 * H :]
            bq, bk, bv = b[:H], b[H : 2 * H], b[2 * H :]
            self.put(l + ".attn.in_proj_weight", np.concatenate([wq[p_qk][:, p_src], wk[p_qk][:, p_tgt], wv[p_v][:, p_tgt]], 0))
            self.put(l + ".attn.in_proj_bias", np.concatenate([bq[p_qk], bk[p_qk], bv[p_v]], 0))
            self.put(l + ".attn.out_proj_weight", self.src[l + ".attn.out_proj_weight"][p_src][:, p_v])
            self.put(l + ".attn.out_proj_bias", self.src[l + ".attn.out_proj_bias"][p_src])
            p_f = self.free()
            self.linear(l + ".linear1", p_f, p_src)
            self.linear(l + ".linear2", p_src, p_f)

    def gru(self, pfx: str, n_layer: int, p_in, p_last=None) -> np.ndarray:
        """nn.GRU naming, gate blocks (r, z, n) of H rows each.  Returns the layout of the top layer's output."""
        p_prev = p_in
        for i in range(n_layer):
            p_h = p_last if (p_last is not None and i == n_layer - 1) else self.free()
            rows = np.concatenate([g * H + p_h for g in range(3)])
            self.put(f"{pfx}.weight_ih_l{i}", self.src[f"{pfx}.weight_ih_l{i}"][rows][:, p_prev])
            self.put(f"{pfx}.weight_hh_l{i}", self.src[f"{pfx}.weight_hh_l{i}"][rows][:, p_h])
            self.put(f"{pfx}.bias_ih_l{i}", self.src[f"{pfx}.bias_ih_l{i}"][rows])
            self.put(f"{pfx}.bias_hh_l{i}", self.src[f"{pfx}.bias_hh_l{i}"][rows])
            p_prev = p_h
        return p_prev


def permute_state_dict(sd: Dict[str, np.ndarray], seed: int):
    """Returns (permuted state_dict, {"agent": p, "tl": p, "map": p}) -- the three layouts the `InputPeEncoder` hooks apply."""
    c = _Ctx(sd, seed)
    p_ag, p_tl, p_map = c.free(), c.free(), c.free()

    # map encoder: node tokens -> densetnt block (residual) -> max pool -> polyline self-attention (residual): one layout
    c.transformer("model.map_encoder.transformer_densetnt", 3, p_map, p_map)
    c.transformer("model.map_encoder.transformer_self_attn", 1, p_map, p_map)

    # policy trunk (traffic_bots.py:205-241); the latent encoder's as2pl / as2tl are aliases of the policy's (shared storage)
    for blk, p_t in (("transformer_as2pl", p_map), ("transformer_as2tl", p_tl)):
        c.transformer(f"model.{blk}", 3, p_ag, p_t)
        for k in [k for k in sd if k.startswith(f"model.latent_encoder.{blk}.")]:
            c.put(k, c.out["model." + k[len("model.latent_encoder."):]])
    c.transformer("model.agent_interaction.transformer", 3, p_ag, p_ag)
    p_x = c.gru("model.agent_temporal.rnn", 3, p_ag)

    # add_goal: u = mlp_in(map_feature[dest]); h = mlp_out([x | u]); x = h + x
    p1, p2, p_u = c.free(), c.free(), c.free()
    c.linear("model.add_goal.mlp_in.fc_layers.0", p1, p_map)
    c.norm("model.add_goal.mlp_in.fc_layers.1", p1)
    c.linear("model.add_goal.mlp_in.fc_layers.4", p2, p1)
    c.norm("model.add_goal.mlp_in.fc_layers.5", p2)
    c.linear("model.add_goal.mlp_in.fc_layers.8", p_u, p2)
    c.norm("model.add_goal.mlp_in.fc_layers.9", p_u)
    p_f = c.free()
    c.linear("model.add_goal.mlp_out.fc_layers.0", p_f, np.concatenate([p_x, H + p_u]))
    c.linear("model.add_goal.mlp_out.fc_layers.3", p_x, p_f)
    # add_latent: z keeps its 16 channels (the stored eps / latent samples refer to them)
    p1, p_u, p_f = c.free(), c.free(), c.free()
    c.linear("model.add_latent.mlp_in.fc_layers.0", p1, None)
    c.linear("model.add_latent.mlp_in.fc_layers.3", p_u, p1)
    c.linear("model.add_latent.mlp_out.fc_layers.0", p_f, np.concatenate([p_x, H + p_u]))
    c.linear("model.add_latent.mlp_out.fc_layers.3", p_x, p_f)
    for i in range(3):
        p_f = c.free()
        c.linear(f"action_head.mlp_mean.{i}.fc_layers.0", p_f, p_x)
        c.linear(f"action_head.mlp_mean.{i}.fc_layers.2", None, p_f)

    # personality encoder (latent_encoder.py:70-147): interaction (tgt = the raw agent feature) -> GRU -> max over time -> dist MLP
    for which in ("prior", "post"):
        c.transformer(f"model.latent_encoder.agent_interaction_{which}.transformer", 3, p_ag, p_ag)
        p_l = c.gru(f"model.latent_encoder.agent_temporal_{which}.rnn", 3, p_ag)
        p_f = c.free()
        c.linear(f"model.latent_encoder.latent_{which}_dist.mlp_mean.fc_layers.0", p_f, p_l)
        c.linear(f"model.latent_encoder.latent_{which}_dist.mlp_mean.fc_layers.2", None, p_f)

    # destination predictor (goal_manager.py:202-333, mode "mlp", res_add_gru: the GRU output is added to the agent feature)
    c.gru("model.goal_manager.goal_predictor.gru_as.rnn", 3, p_ag, p_last=p_ag)
    p1, p2 = c.free(), c.free()
    g = "model.goal_manager.goal_predictor.mlp.fc_layers"
    c.linear(f"{g}.0", p1, np.concatenate([p_map, H + p_ag]))
    c.norm(f"{g}.1", p1)
    c.linear(f"{g}.3", p2, p1)
    c.norm(f"{g}.4", p2)
    c.linear(f"{g}.6", None, p2)

    # everything else (token-encoder MLPs on attr inputs, log_std vectors, buffers) has no hidden-channel axis to re-label
    untouched = [k for k in sd if k not in c.done]
    for k in untouched:
        assert (".mlp.fc_layers." in k and "_encoder" in k) or "log_std" in k or k.startswith("pre_processing."), \
            f"channel_perm: no rule for {k} {sd[k].shape}"
        c.out[k] = sd[k]
    return {k: c.out[k] for k in sd}, {"agent": p_ag, "tl": p_tl, "map": p_map}


def install_hooks(model, layouts) -> list:
    """Forward hooks on the reference's three InputPeEncoder instances (`model` = the reference WaymoMotion): feature[..., p]."""
    import torch

    handles = []
    for mod, key in ((model.model.agent_encoder, "agent"), (model.model.tl_encoder, "tl"), (model.model.map_encoder.input_pe_encoder, "map")):
        idx = torch.from_numpy(np.asarray(layouts[key]).astype(np.int64))
        handles.append(mod.register_forward_hook(lambda m, a, out, _i=idx: out.index_select(-1, _i)))
    return handles
