"""Host-side mirror of the reference's `loss/losses_duett.py`: `VanillaKLKD`, `StudentKDLoss`, `PathologyMultiLabelLoss`,
`DualPathologyLoss`, `build_kd_loss` — same constructor arguments, buffers and output dicts; value + gradient come from
one HIP launch each (`medp_dual_pathology_loss`, `medp_student_kd_loss`)."""
from __future__ import annotations

import torch
import torch.nn as nn

from . import autograd_ops as A


class VanillaKLKD(nn.Module):
    """losses_duett.py:8-25: T^2 * mean KL(sigmoid(z_t/T) || sigmoid(z_s/T)), probabilities clamped to [eps, 1-eps]."""

    def __init__(self, T: float = 4.0, eps: float = 1e-7):
        super().__init__()
        self.T = T
        self.eps = eps
        if eps != 1e-7:
            raise ValueError("the HIP KD kernel is built for the reference's eps = 1e-7")

    def forward(self, z_s: torch.Tensor, z_t: torch.Tensor) -> torch.Tensor:
        out = A.StudentKDLossFn.apply(z_s, z_t.detach(), torch.zeros_like(z_s), float(self.T), 0.0, 1.0)   # alpha = 0 -> KD only
        return out[2] if not z_s.requires_grad else out[0]


KD_LOSSES = {"vanilla_kl": VanillaKLKD}


def build_kd_loss(name: str, **kwargs) -> nn.Module:
    if name not in KD_LOSSES:
        raise ValueError(f"unknown KD loss: {name!r}. available: {list(KD_LOSSES)}")
    return KD_LOSSES[name](**kwargs)


class StudentKDLoss(nn.Module):
    """losses_duett.py:39-57: total = alpha * BCE(z_s, y) + (1 - alpha) * L_kd(z_s, z_t)."""

    def __init__(self, kd_name: str = "vanilla_kl", kd_T: float = 4.0, kd_alpha: float = 0.5, pos_weight: float | None = None):
        super().__init__()
        self.alpha = kd_alpha
        self.kd = build_kd_loss(kd_name, T=kd_T)
        self.pos_weight = pos_weight

    def forward(self, z_s: torch.Tensor, z_t: torch.Tensor, y: torch.Tensor) -> dict:
        out = A.StudentKDLossFn.apply(z_s, z_t.detach(), y.float(), float(self.kd.T), float(self.alpha),
                                      1.0 if self.pos_weight is None else float(self.pos_weight))
        return {"total": out[0], "bce": out[1].detach(), "kd": out[2].detach()}


class DualPathologyLoss(nn.Module):
    """losses_duett.py:131-194: three branches x K masked per-label BCE means, label-weighted, alpha-weighted."""

    def __init__(self, label_weights: torch.Tensor, pos_weight: torch.Tensor | None = None, alpha_img: float = 0.5,
                 alpha_ts: float = 0.5, alpha_fus: float = 1.0, eps: float = 1e-6):
        super().__init__()
        self.register_buffer("label_weights", label_weights.float())
        if pos_weight is not None:
            self.register_buffer("pos_weight", pos_weight.float())
        else:
            self.pos_weight = None
        self.alpha_img, self.alpha_ts, self.alpha_fus = float(alpha_img), float(alpha_ts), float(alpha_fus)
        self.eps = eps
        self.n_pathologies = int(label_weights.numel())

    def forward(self, img_logits, ts_logits, fusion_logits, y_multi, y_multi_mask) -> dict:
        K = self.n_pathologies
        out = A.DualPathologyLossFn.apply(img_logits, ts_logits, fusion_logits, y_multi, y_multi_mask, self.label_weights,
                                          self.pos_weight, self.alpha_img, self.alpha_ts, self.alpha_fus, float(self.eps))
        d = out.detach()
        return {"total": out[0], "img_total": d[1], "ts_total": d[2], "fus_total": d[3], "img_per": d[4:4 + K],
                "ts_per": d[4 + K:4 + 2 * K], "fus_per": d[4 + 2 * K:4 + 3 * K]}


class PathologyMultiLabelLoss(nn.Module):
    """losses_duett.py:63-125 (two-stage variant): the same per-label masked BCE over stage2 / stage4 logits."""

    def __init__(self, label_weights: torch.Tensor, pos_weight: torch.Tensor | None = None, alpha_stage2: float = 0.5,
                 alpha_stage4: float = 1.0, eps: float = 1e-6):
        super().__init__()
        self.register_buffer("label_weights", label_weights.float())
        if pos_weight is not None:
            self.register_buffer("pos_weight", pos_weight.float())
        else:
            self.pos_weight = None
        self.alpha_stage2, self.alpha_stage4 = float(alpha_stage2), float(alpha_stage4)
        self.eps = eps
        self.n_pathologies = int(label_weights.numel())

    def forward(self, stage2_logits, stage4_logits, y_multi, y_multi_mask) -> dict:
        K = self.n_pathologies
        out = A.DualPathologyLossFn.apply(stage2_logits, stage4_logits, stage4_logits.detach(), y_multi, y_multi_mask,
                                          self.label_weights, self.pos_weight, self.alpha_stage2, self.alpha_stage4, 0.0, float(self.eps))
        d = out.detach()
        return {"total": out[0], "stage2_total": d[1], "stage4_total": d[2], "stage2_per": d[4:4 + K],
                "stage4_per": d[4 + K:4 + 2 * K]}
