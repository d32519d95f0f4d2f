"""fp32 restatement of the loss side (SURVEY.md §8a rows a12–a16):
`loss/losses_duett.py:8-194`, `training_duett/engine.py:149-165,217-237`,
`cxr_linear_training.ipynb:426-437`."""
from __future__ import annotations

import torch
import torch.nn.functional as F


def per_pathology_bce(logits, y, mask, pos_weight=None, eps=1e-6):
    """losses_duett.py:152-164 — per label k: Σ(bce·m)/(Σm + eps)."""
    pw = None if pos_weight is None else pos_weight.unsqueeze(0)
    l = F.binary_cross_entropy_with_logits(logits, y, reduction="none", pos_weight=pw)
    return (l * mask).sum(0) / (mask.sum(0) + eps)


def dual_pathology_loss(img, ts, fus, y, mask, label_weights, pos_weight=None,
                        alpha_img=0.5, alpha_ts=0.5, alpha_fus=1.0, eps=1e-6):
    """losses_duett.py:166-194."""
    ip = per_pathology_bce(img, y, mask, pos_weight, eps)
    tp = per_pathology_bce(ts, y, mask, pos_weight, eps)
    fp = per_pathology_bce(fus, y, mask, pos_weight, eps)
    it, tt, ft = (label_weights * ip).sum(), (label_weights * tp).sum(), (label_weights * fp).sum()
    total = alpha_img * it + alpha_ts * tt + alpha_fus * ft
    return {"total": total, "img_total": it.detach(), "ts_total": tt.detach(), "fus_total": ft.detach(),
            "img_per": ip.detach(), "ts_per": tp.detach(), "fus_per": fp.detach()}


def pathology_multilabel_loss(s2, s4, y, mask, label_weights, pos_weight=None, alpha_stage2=0.5,
                              alpha_stage4=1.0, eps=1e-6):
    """losses_duett.py:107-125."""
    p2 = per_pathology_bce(s2, y, mask, pos_weight, eps)
    p4 = per_pathology_bce(s4, y, mask, pos_weight, eps)
    t2, t4 = (label_weights * p2).sum(), (label_weights * p4).sum()
    return {"total": alpha_stage2 * t2 + alpha_stage4 * t4, "stage2_total": t2.detach(),
            "stage4_total": t4.detach(), "stage2_per": p2.detach(), "stage4_per": p4.detach()}


def vanilla_kl_kd(z_s, z_t, T=4.0, eps=1e-7):
    """losses_duett.py:20-25."""
    p_t = torch.sigmoid(z_t.detach() / T).clamp(eps, 1 - eps)
    p_s = torch.sigmoid(z_s / T).clamp(eps, 1 - eps)
    kl = p_t * (p_t.log() - p_s.log()) + (1 - p_t) * ((1 - p_t).log() - (1 - p_s).log())
    return (T ** 2) * kl.mean()


def student_kd_loss(z_s, z_t, y, T=4.0, alpha=0.5, pos_weight=None):
    """losses_duett.py:53-57."""
    kd = vanilla_kl_kd(z_s, z_t, T)
    pw = None if pos_weight is None else torch.tensor([pos_weight], dtype=torch.float32)
    bce = F.binary_cross_entropy_with_logits(z_s, y.float(), pos_weight=pw)
    return {"total": alpha * bce + (1 - alpha) * kd, "bce": bce.detach(), "kd": kd.detach()}


def aux_residual_kl(img_logits, scaled_correction, y_multi, mask, eps=0.05):
    """engine.py:149-165."""
    y = y_multi.float()
    ys = y * (1 - eps) + (1 - y) * eps
    p = torch.sigmoid(img_logits.detach() + scaled_correction).clamp(min=1e-6, max=1 - 1e-6)
    kl = ys * (torch.log(ys) - torch.log(p)) + (1 - ys) * (torch.log(1 - ys) - torch.log(1 - p))
    m = mask.float()
    return (kl * m).sum() / m.sum().clamp(min=1.0)


def lp_regularisers(beta, scaled_correction, beta_l2, corr_l2):
    """engine.py:217-223."""
    return beta_l2 * (beta ** 2).mean(), corr_l2 * (scaled_correction ** 2).mean()


def masked_bce_global(logits, y, mask):
    """cxr_linear_training.ipynb:426-437 — one global masked mean (config 2)."""
    l = F.binary_cross_entropy_with_logits(logits, y, reduction="none")
    return (l * mask).sum() / mask.sum().clamp(min=1.0)
