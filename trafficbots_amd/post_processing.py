"""Host mirror of the reference's `WaymoPostProcessing` (`src/data_modules/waymo_post_processing.py:8-81`): same constructor
arguments (config group `waymo_post_processing`, `configs/model/traffic_bots.yaml:179-186`), same `forward` signature and
result dict; the work runs in `tb_post_process` (`trafficbots_amd/csrc/tb_post_kernels.hip`)."""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Sequence

import torch
from torch import Tensor

from . import hip
from .runtime import HipEngine, as_bool, as_u8


class WaymoPostProcessing:
    def __init__(self, engine: HipEngine, k_pred: int = 6, score_temperature: float = 1e2, mpa_nms_thresh: Sequence[float] = (),
                 mtr_nms_thresh: Sequence[float] = (), aggr_thresh: Sequence[float] = (), n_iter_em: int = 3,
                 use_ade: bool = True) -> None:
        self.engine = engine
        self.k_pred = int(k_pred)
        self.score_temperature = float(score_temperature)
        self.mpa_nms_thresh = [float(x) for x in mpa_nms_thresh]
        self.mtr_nms_thresh = [float(x) for x in mtr_nms_thresh]
        self.n_iter_em = int(n_iter_em)
        self.use_ade = bool(use_ade)
        if len(list(aggr_thresh)) > 0:
            # the reference itself cannot run this branch: traj_aggr compares a Tensor with a python list (:231, TypeError)
            raise NotImplementedError("waymo_post_processing.aggr_thresh: traj_aggr is not runnable in the reference and not built")
        for name, th in (("mpa_nms_thresh", self.mpa_nms_thresh), ("mtr_nms_thresh", self.mtr_nms_thresh)):
            if len(th) not in (0, 3):
                raise ValueError(f"waymo_post_processing.{name} must be [] or [veh, ped, cyc]")

    def __call__(self, valid: Tensor, scores: Tensor, trajs: Tensor, agent_type: Tensor) -> Dict[str, Optional[Tensor]]:
        return self.forward(valid, scores, trajs, agent_type)

    def forward(self, valid: Tensor, scores: Tensor, trajs: Tensor, agent_type: Tensor) -> Dict[str, Optional[Tensor]]:
        """valid [B,A] bool, scores [B,A,NP] (not normalised), trajs [B,A,NP,S,2..4], agent_type [B,A,3] one-hot (or [B,A] index).
        Returns waymo_valid [B,S,A], waymo_trajs [B,S,A,K,2], waymo_scores [B,A,K], waymo_yaw_bbox / waymo_spd [B,S,A,K,1] or None
        (`waymo_post_processing.py:39-46`), plus `mode_idx` [B,A,K]: which input mode every output mode is."""
        eng, dev = self.engine, self.engine.device
        b, a, n_pred, n_step, d = trajs.shape
        k = min(self.k_pred, n_pred)
        f32 = torch.float32
        valid_u8 = as_u8(valid.to(dev))
        scores_c = scores.to(dev).to(f32).contiguous()
        # the trajectories are read where they lie when the D components are adjacent (a rollout buffer's [B*K,A,S_all,4] viewed as
        # [B,A,K,S,4] from the first future step on is such a view: tb_post_io.traj_strides) -- no re-laid-out copy
        trajs_c = trajs.to(dev).to(f32)
        strided = not trajs_c.is_contiguous()
        if strided and trajs_c.stride(4) != 1:
            trajs_c, strided = trajs_c.contiguous(), False
        ty = agent_type.to(dev)
        ty = (ty.to(torch.int32).argmax(-1) if ty.dim() == 3 else ty).to(torch.int32).contiguous()
        out = {
            "waymo_trajs": torch.empty(b, n_step, a, k, 2, device=dev, dtype=f32),
            "waymo_yaw_bbox": torch.empty(b, n_step, a, k, 1, device=dev, dtype=f32) if d >= 3 else None,
            "waymo_spd": torch.empty(b, n_step, a, k, 1, device=dev, dtype=f32) if d >= 4 else None,
            "waymo_scores": torch.empty(b, a, k, device=dev, dtype=f32),
            "waymo_valid": torch.empty(b, n_step, a, device=dev, dtype=torch.uint8),
            "mode_idx": torch.empty(b, a, k, device=dev, dtype=torch.int32),
        }
        io = hip.TbPostIO()
        io.n_scene, io.n_agent, io.n_pred, io.n_step, io.d_traj = b, a, n_pred, n_step, d
        io.k_pred, io.score_temperature, io.use_ade = self.k_pred, self.score_temperature, int(self.use_ade)
        io.n_mpa, io.n_mtr = len(self.mpa_nms_thresh), len(self.mtr_nms_thresh)
        for i, v in enumerate(self.mpa_nms_thresh):
            io.mpa_nms_thresh[i] = v
        for i, v in enumerate(self.mtr_nms_thresh):
            io.mtr_nms_thresh[i] = v
        io.valid, io.scores = hip.ptr(valid_u8, hip.c_u8p), hip.ptr(scores_c, hip.c_f32p)
        if strided:
            for i in range(4):
                io.traj_strides[i] = trajs_c.stride(i)
            base = trajs_c.as_strided((trajs_c.untyped_storage().nbytes() // 4 - trajs_c.storage_offset(),), (1,))  # (first element .. end of the storage)
            io.trajs = hip.ptr(base, hip.c_f32p)
        else:
            io.trajs = hip.ptr(trajs_c, hip.c_f32p)
        io.agent_type = hip.ptr(ty, hip.c_i32p)
        io.waymo_trajs = hip.ptr(out["waymo_trajs"], hip.c_f32p)
        io.waymo_yaw_bbox = hip.ptr(out["waymo_yaw_bbox"], hip.c_f32p)
        io.waymo_spd = hip.ptr(out["waymo_spd"], hip.c_f32p)
        io.waymo_scores = hip.ptr(out["waymo_scores"], hip.c_f32p)
        io.waymo_valid = hip.ptr(out["waymo_valid"], hip.c_u8p)
        io.mode_idx = hip.ptr(out["mode_idx"], hip.c_i32p)
        eng._check(eng.lib.tb_post_process(eng._ctx, C.byref(io), eng._stream()), "tb_post_process")
        out["waymo_valid"] = as_bool(out["waymo_valid"])
        out["_keepalive"] = (valid_u8, scores_c, trajs_c, ty)
        return out
