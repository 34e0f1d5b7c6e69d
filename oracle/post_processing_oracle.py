"""CPU restatement of the reference's WaymoPostProcessing -- TEST INFRASTRUCTURE ONLY.

Follows `src/data_modules/waymo_post_processing.py` of zhejz/TrafficBots: `forward` (:33-81), `mpa_nms` (:83-121),
`mtr_nms` (:123-170), `traj_topk` (:172-192); `traj_aggr` (:194-295) is not runnable in the reference (TypeError at :231).  Pinned by tests/golden/post_processing.npz, which
holds what the imported reference returned on seeded synthetic (valid, scores, trajs, agent_type) inputs for five
configurations (tools/gen_golden_post.py).  Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg
may import this module.

Written per (scene, agent) with explicit loops -- the form the HIP kernel `k_post_process` has -- instead of the reference's
batched tensor ops.
"""
from __future__ import annotations

from typing import Dict, List

import numpy as np


def _pair_dist(xy_a: np.ndarray, xy_b: np.ndarray, use_ade: bool) -> np.float32:
    """distance of two [S,2] trajectories: mean over steps of the point distances, or the final-point distance."""
    d = np.sqrt(((xy_a - xy_b) ** 2).sum(-1, dtype=np.float32), dtype=np.float32)
    return np.float32(d.mean(dtype=np.float32)) if use_ade else np.float32(d[-1])


def post_process(valid: np.ndarray, scores: np.ndarray, trajs: np.ndarray, agent_type: np.ndarray, k_pred: int,
                 score_temperature: float, mpa_nms_thresh: List[float], mtr_nms_thresh: List[float], aggr_thresh: List[float],
                 n_iter_em: int, use_ade: bool) -> Dict[str, np.ndarray]:
    """valid [B,A] bool, scores [B,A,NP] (un-normalised), trajs [B,A,NP,S,D], agent_type [B,A,3] bool.
    Returns waymo_trajs [B,S,A,K,2], waymo_yaw_bbox / waymo_spd [B,S,A,K,1] (or None), waymo_scores [B,A,K], waymo_valid [B,S,A]
    and mode_idx [B,A,K] (the selected modes)."""
    f32 = np.float32
    b_, a_, n_pred, n_step, d_traj = trajs.shape
    k_out = k_pred if n_pred > k_pred else n_pred
    out_t = np.zeros((b_, a_, k_out, n_step, d_traj), f32)
    out_s = np.zeros((b_, a_, k_out), f32)
    out_i = np.full((b_, a_, k_out), -1, np.int64)
    for b in range(b_):
        for a in range(a_):
            tr = trajs[b, a].astype(f32)
            sc = scores[b, a].astype(f32)
            sc = sc / sc.sum(dtype=f32)
            ty = agent_type[b, a]

            def type_thresh(th):
                t = f32(0)
                for i in range(len(th)):
                    t = f32(t + f32(ty[i]) * f32(th[i]))
                return t

            def within(th):
                w = np.zeros((n_pred, n_pred), bool)
                for i in range(n_pred):
                    for j in range(n_pred):
                        w[i, j] = _pair_dist(tr[i, :, :2], tr[j, :, :2], use_ade) < th
                return w

            if n_pred > k_pred:
                if len(aggr_thresh) > 0:
                    # traj_aggr (:194-295) cannot run in the reference: `Tensor < list` at :231 raises TypeError
                    raise NotImplementedError("aggr_thresh: the reference's traj_aggr raises TypeError for any non-empty list")
                elif len(mtr_nms_thresh) > 0:  # mtr_nms (:123-170)
                    w = within(type_thresh(mtr_nms_thresh))
                    clone = sc.copy()
                    idx = []
                    for _ in range(k_pred):
                        i = int(clone.argmax())
                        clone = clone * (np.where(w[i], f32(0), f32(1)) * f32(0.99) + f32(0.01)).astype(f32)
                        clone[i] = f32(-1)
                        idx.append(i)
                    sel_t, sel_s, sel_i = tr[idx], sc[idx] / sc[idx].sum(dtype=f32), idx
                else:  # traj_topk (:172-192); the reference's order inside the k is unspecified (sorted=False)
                    idx = sorted(np.argsort(-sc, kind="stable")[:k_pred].tolist())
                    sel_t, sel_s, sel_i = tr[idx], sc[idx] / sc[idx].sum(dtype=f32), idx
            else:
                sel_t, sel_s, sel_i = tr, sc, list(range(n_pred))
            sel_s = np.array(sel_s, f32)
            if len(mpa_nms_thresh) > 0:  # mpa_nms (:83-121), on the selected modes
                th = type_thresh(mpa_nms_thresh)
                k_now = sel_t.shape[0]
                if valid[b, a]:
                    w = np.zeros((k_now, k_now), bool)
                    for i in range(k_now):
                        for j in range(k_now):
                            w[i, j] = _pair_dist(sel_t[i, :, :2], sel_t[j, :, :2], use_ade) < th
                    for kk in np.argsort(-sel_s, kind="stable"):
                        if (w[kk] & (sel_s > sel_s[kk])).any():
                            sel_s[kk] = f32(1e-3)
                sel_s = sel_s / sel_s.sum(dtype=f32)
            if score_temperature > 0:
                z = np.log(sel_s).astype(f32) / f32(score_temperature)
                e = np.exp(z - z.max(), dtype=f32)
                sel_s = (e / e.sum(dtype=f32)).astype(f32)
            out_t[b, a], out_s[b, a], out_i[b, a] = sel_t, sel_s, sel_i
    moved = np.moveaxis(out_t, 3, 1)  # [B,S,A,K,D]
    return {
        "waymo_trajs": moved[..., :2].copy(),
        "waymo_yaw_bbox": moved[..., 2:3].copy() if d_traj >= 3 else None,
        "waymo_spd": moved[..., 3:4].copy() if d_traj >= 4 else None,
        "waymo_scores": out_s,
        "waymo_valid": np.broadcast_to(valid[:, None, :], (b_, n_step, a_)).copy(),
        "mode_idx": out_i,
    }
