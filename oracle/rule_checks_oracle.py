"""CPU restatement of the reference's flag-gated traffic-rule checks -- TEST INFRASTRUCTURE ONLY.

Follows `src/utils/traffic_rule_checker.py` of zhejz/TrafficBots: `_check_collided` (:122-163), `_check_run_road_edge`
(:165-203), `_check_run_red_light` (:205-262), `_check_passive` (:264-335), `_get_agent_bbox` (:518-543),
`_get_road_edge` / `_get_lane_center` (:557-592), `ccw` (:595-596) and the accumulation in `check` (:412-516).
Pinned by tests/golden/rules_k2.npz and rules_passive.npz, which hold what the imported reference returned together
with the (valid, state) pairs it handed to `TrafficRuleChecker.check` (tools/gen_golden.py).

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg may import this module.

The checks are pure functions of the per-step post-override simulator state, so they are evaluated over a recorded
trajectory: state [N, A, S, 4] (x, y, yaw, speed), valid [N, A, S]; instance n belongs to scene n // K.
"""
from __future__ import annotations

from typing import Dict

import torch
from torch import Tensor

COLLISION_SIZE_SCALE = 1.1  # traffic_rule_checker.py:28-30


def _corners(state: Tensor, size_lw: Tensor) -> Tensor:
    """[..., 4] states + [..., 2] (length, width) -> [..., 4 corners, 2]  (_get_agent_bbox, :518-543)."""
    c, s = torch.cos(state[..., 2]), torch.sin(state[..., 2])
    fwd = torch.stack([c, s], -1)
    rgt = torch.stack([s, -c], -1)
    of = 0.5 * size_lw[..., [0]] * fwd
    orr = 0.5 * size_lw[..., [1]] * rgt
    off = torch.stack([-of + orr, of + orr, of - orr, -of - orr], dim=-2)
    return state[..., None, :2] + off


def _ccw(a: Tensor, b: Tensor, c: Tensor) -> Tensor:
    return (c[..., 1] - a[..., 1]) * (b[..., 0] - a[..., 0]) > (b[..., 1] - a[..., 1]) * (c[..., 0] - a[..., 0])


def rule_checks(state: Tensor, valid: Tensor, k_futures: int, step_start: int, agent_type: Tensor, agent_size: Tensor,
                map_valid: Tensor, map_type: Tensor, map_pos: Tensor, map_dir: Tensor, tl_valid: Tensor, tl_pos: Tensor,
                tl_state: Tensor, enable_collided: bool = True, enable_run_road_edge: bool = True,
                enable_run_red_light: bool = True, enable_passive: bool = True) -> Dict[str, Tensor]:
    """state [N,A,S,4], valid [N,A,S] bool; scene tensors un-repeated:
    agent_type [B,A,3] bool, agent_size [B,A,3], map_valid [B,P,20], map_type [B,P,11] bool, map_pos/map_dir [B,P,20,2],
    tl_valid [B,NH,T], tl_pos [B,NH,T,2], tl_state [B,NH,T,5] bool.  Returns the eight [N,A,S] bool arrays of `check`."""
    n, a, s_len, _ = state.shape
    rep = lambda x: x.repeat_interleave(k_futures, 0)  # noqa: E731
    veh = rep(agent_type[:, :, 0])
    pedcyc = rep(agent_type[:, :, 1])
    size = rep(agent_size[..., :2]) * COLLISION_SIZE_SCALE
    zeros = torch.zeros(n, a, s_len, dtype=torch.bool)
    out = {k: zeros.clone() for k in ("collided_this_step", "run_road_edge_this_step", "run_red_light_this_step", "passive_this_step")}

    eye = torch.eye(a, dtype=torch.bool)[None]
    coll_invalid = eye | (pedcyc[:, None, :] & pedcyc[:, :, None])  # :56-61
    edge_ok = rep((map_valid & map_type[:, :, [4, 5, 7]].any(-1, keepdim=True)).flatten(1, 2))  # :571-573
    edge_a = rep(map_pos.flatten(1, 2))
    edge_b = rep((map_pos + map_dir).flatten(1, 2))
    lane_ok = rep((map_valid & map_type[:, :, :3].any(-1, keepdim=True)).flatten(1, 2))  # :587-589
    lane_xy = rep(map_pos.flatten(1, 2))
    red_len = rep(agent_size[:, :, [0]]) * 0.5 * 0.6  # :68-69
    red_wid = rep(agent_size[:, :, [1]]) * 0.5 * 1.8
    n_tl_step = tl_valid.shape[1]
    counter = torch.zeros(n, a, dtype=torch.float32)

    for si in range(s_len):
        st, va = state[:, :, si], valid[:, :, si]
        step = step_start + si
        box = _corners(st, size)  # [N,A,4,2]
        nxt = box.roll(-1, dims=2)
        if enable_collided:
            line = torch.cat([nxt[..., [1]] - box[..., [1]], box[..., [0]] - nxt[..., [0]],
                              nxt[..., [0]] * box[..., [1]] - nxt[..., [1]] * box[..., [0]]], -1)  # [N,A,4,3]
            pts = torch.cat([box, torch.ones_like(box[..., [0]])], -1)  # [N,A,4,3]
            prod = line[:, :, None, :, None, :] * pts[:, None, :, None, :, :]  # [N,Ai,Aj,4 lines,4 points,3]
            outside = prod.sum(-1) > 0
            sep = outside.all(-1).any(-1)  # a line of i with all corners of j on its outer side
            sep = sep | sep.transpose(1, 2)
            free = sep | coll_invalid | ~(va[:, :, None] & va[:, None, :])
            out["collided_this_step"][:, :, si] = ~free.all(-1)
        if enable_run_road_edge:
            p0, p1 = box[:, :, None, :, :], nxt[:, :, None, :, :]  # [N,A,1,4,2]
            q0, q1 = edge_a[:, None, :, None, :], edge_b[:, None, :, None, :]  # [N,1,E,1,2]
            cross = (_ccw(p0, q0, q1) != _ccw(p1, q0, q1)) & (_ccw(p0, p1, q0) != _ccw(p0, p1, q1))  # [N,A,E,4]
            hit = (cross.any(-1) & edge_ok[:, None, :]).any(-1)
            out["run_road_edge_this_step"][:, :, si] = hit & va & veh
        tls = min(step, n_tl_step - 1)  # :448
        tv, tp, ts = rep(tl_valid[:, tls]), rep(tl_pos[:, tls]), rep(tl_state[:, tls])
        c, s = torch.cos(st[..., 2]), torch.sin(st[..., 2])
        fwd = torch.stack([c, s], -1)[:, :, None, :]
        rgt = torch.stack([s, -c], -1)[:, :, None, :]
        if enable_run_red_light:
            xy0 = st[..., :2][:, :, None, :]
            xy1 = xy0 + 0.1 * st[..., [3]][:, :, None, :] * fwd
            d0, d1 = tp[:, None] - xy0, tp[:, None] - xy1
            in0 = ((d0 * fwd).sum(-1).abs() < red_len) & ((d0 * rgt).sum(-1).abs() < red_wid)
            in1 = ((d1 * fwd).sum(-1).abs() < red_len) & ((d1 * rgt).sum(-1).abs() < red_wid)
            ok = (va & veh)[:, :, None] & (tv & ts[:, :, 1])[:, None, :]
            out["run_red_light_this_step"][:, :, si] = (in0 & ~in1 & ok).any(-1)
        if enable_passive:
            near = (torch.norm(st[:, :, None, :2] - lane_xy[:, None], dim=-1) < 2) & lane_ok[:, None]
            near = near.any(-1)
            slow = st[..., 3] < 5
            tmask = (tv & ts[:, :, [0, 1, 2, 4]].any(-1))[:, None]
            tvec = tp[:, None] - st[:, :, None, :2]
            tnorm = torch.norm(tvec, dim=-1)
            red_ahead = ((tnorm < 10) & (((fwd * tvec).sum(-1) / tnorm) > 0.95) & tmask).any(-1)
            avec = st[:, None, :, :2] - st[:, :, None, :2]
            anorm = torch.norm(avec, dim=-1)
            ahead = ((anorm < 10) & (((fwd * avec).sum(-1) / anorm) > 0.95) & va[:, None, :] & va[:, :, None] & ~eye).any(-1)
            raw = va & veh & near & slow & ~red_ahead & ~ahead
            counter = (counter + raw) * raw  # :331-333
            out["passive_this_step"][:, :, si] = counter > 20

    res = {}
    for name in ("collided", "run_road_edge", "run_red_light", "passive"):
        this = out[f"{name}_this_step"]
        res[f"{name}_this_step"] = this
        res[name] = torch.cummax(this.to(torch.uint8), dim=2).values.bool()  # self.x = self.x | x_this_step
    return res
